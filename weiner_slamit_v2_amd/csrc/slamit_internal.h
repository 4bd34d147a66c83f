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


// One pinned staging block, one device slab and one stream per host thread and call site, kept between calls: the
// per-frame entry points (pose, Sim3, guided search ...) are called every frame by a tracking thread, and a fresh
// hipMalloc / hipFree pair per call costs more than their kernels.
struct SlamitScratch { int device; unsigned char* host; size_t host_bytes; unsigned char* dev; size_t dev_bytes; hipStream_t st; };
inline hipError_t slamit_scratch_reserve(SlamitScratch& S, int device, size_t bytes) {
    if (S.device == device && S.host_bytes >= bytes && S.dev_bytes >= bytes && S.st) return hipSuccess;
    if (S.st) hipStreamSynchronize(S.st);
    if (S.host) hipHostFree(S.host);
    if (S.dev) hipFree(S.dev);
    if (S.st && S.device != device) { hipStreamDestroy(S.st); S.st = nullptr; }   // a stream belongs to the device it was created on
    S.host = nullptr; S.dev = nullptr; S.host_bytes = S.dev_bytes = 0; S.device = device;
    hipError_t e = hipSuccess;
    if (!S.st) e = hipStreamCreateWithFlags(&S.st, hipStreamNonBlocking);   // the caller has made `device` current
    const size_t want = bytes + bytes / 2 + 4096;
    if (e == hipSuccess) e = hipHostMalloc((void**)&S.host, want, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&S.dev, want);
    if (e == hipSuccess) S.host_bytes = S.dev_bytes = want;
    return e;
}

#endif
