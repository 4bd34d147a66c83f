// slamit_misc.hip — error string, version, device count (include/slamit.h "misc").
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "../../include/slamit.h"
#include "slamit_internal.h"

#include <vector>

static thread_local char g_err[512] = "";
static thread_local int g_device = -1;                       // slamit_set_device() of this thread; -1: the current device
static thread_local std::vector<SlamitScratch*>* g_scratch = nullptr;   // this thread's staging blocks (leaked pointer list: a few words)

int slamit_default_device() {
    if (g_device >= 0) return g_device;
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) { (void)hipGetLastError(); d = 0; }
    return d;
}

void slamit_scratch_register(SlamitScratch* s) {
    if (!g_scratch) g_scratch = new std::vector<SlamitScratch*>();
    g_scratch->push_back(s);
}

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

int slamit_set_device(int device) {
    int n = 0;
    if (device < -1 || (device >= 0 && (hipGetDeviceCount(&n) != hipSuccess || device >= n))) return slamit_fail(SLAMIT_ERR_ARG, "slamit_set_device: no such device");
    g_device = device;
    return SLAMIT_OK;
}

void slamit_release_thread_scratch(void) {
    if (!g_scratch) return;
    for (SlamitScratch* s : *g_scratch) s->release();
}

int slamit_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

}  // extern "C"
