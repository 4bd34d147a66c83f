// slamit_misc.hip — error string, version, device count (include/slamit.h "misc").
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "../../include/slamit.h"
#include "slamit_internal.h"

static thread_local char g_err[512] = "";

int slamit_fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int slamit_fail_hip(hipError_t e, const char* where) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    (void)hipGetLastError();  // clear the sticky error
    return SLAMIT_ERR_DEVICE;
}

extern "C" {

const char* slamit_last_error(void) { return g_err; }

const char* slamit_version(void) { return "slamit-hip 0.1 (gfx950)"; }

int slamit_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

}  // extern "C"
