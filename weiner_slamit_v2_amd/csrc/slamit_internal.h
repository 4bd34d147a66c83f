// slamit_internal.h — error plumbing shared by the C-ABI translation units.
#ifndef SLAMIT_INTERNAL_H
#define SLAMIT_INTERNAL_H
#include <hip/hip_runtime.h>

int slamit_fail(int code, const char* msg);                 // records msg, returns code
int slamit_fail_hip(hipError_t e, const char* where);       // records "<where>: <hip error>", returns SLAMIT_ERR_DEVICE

#define HIP_TRY(expr)                                                        \
    do {                                                                     \
        hipError_t _e = (expr);                                              \
        if (_e != hipSuccess) return slamit_fail_hip(_e, #expr);             \
    } while (0)

#endif
