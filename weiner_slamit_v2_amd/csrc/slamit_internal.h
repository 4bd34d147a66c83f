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


// Every entry point works on the device it is given and leaves the caller's current device as it found it (a torch
// host thread must not have its device changed under it).
struct SlamitDeviceGuard {
    int prev;
    hipError_t err;
    explicit SlamitDeviceGuard(int device) : prev(-1), err(hipSuccess) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (device != prev) err = hipSetDevice(device);
    }
    ~SlamitDeviceGuard() { int cur = -1; if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) hipSetDevice(prev); }
    SlamitDeviceGuard(const SlamitDeviceGuard&) = delete;
    SlamitDeviceGuard& operator=(const SlamitDeviceGuard&) = delete;
};
#define SLAMIT_USE_DEVICE(device)                                                  \
    SlamitDeviceGuard slamit_device_guard_(device);                                \
    if (slamit_device_guard_.err != hipSuccess) return slamit_fail_hip(slamit_device_guard_.err, "hipSetDevice")

int slamit_default_device();   // slamit_set_device() of this thread, or the current device

// One pinned staging block, one device slab and one stream per host thread and call site, kept between calls: the
// per-frame entry points (pose, Sim3, guided search ...) are called every frame by a tracking thread, and a fresh
// hipMalloc / hipFree pair per call costs more than their kernels.  The blocks are released when the thread exits
// (or by slamit_release_thread_scratch(), which every call site registers with).
struct SlamitScratch {
    int device = -1;
    unsigned char* host = nullptr; size_t host_bytes = 0;
    unsigned char* dev = nullptr; size_t dev_bytes = 0;
    hipStream_t st = nullptr;
    void release() {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) { host = nullptr; dev = nullptr; st = nullptr; host_bytes = dev_bytes = 0; return; }   // runtime already gone (process exit)
        if (st) hipStreamSynchronize(st);
        if (host) hipHostFree(host);
        if (dev) hipFree(dev);
        if (st) hipStreamDestroy(st);
        host = nullptr; dev = nullptr; st = nullptr; host_bytes = dev_bytes = 0; device = -1;
    }
    ~SlamitScratch() { release(); }
};
void slamit_scratch_register(SlamitScratch* s);   // so that slamit_release_thread_scratch() finds it

// the caller has made `device` current; host_bytes of pinned memory, dev_bytes of device memory
inline hipError_t slamit_scratch_reserve(SlamitScratch& S, int device, size_t host_bytes, size_t dev_bytes) {
    if (S.device == device && S.host_bytes >= host_bytes && S.dev_bytes >= dev_bytes && S.st) return hipSuccess;
    if (S.device == -1 && !S.st && !S.host && !S.dev) slamit_scratch_register(&S);
    if (S.st) hipStreamSynchronize(S.st);
    if (S.host) hipHostFree(S.host);
    if (S.dev) hipFree(S.dev);
    if (S.st && S.device != device) { hipStreamDestroy(S.st); S.st = nullptr; }   // a stream belongs to the device it was created on
    S.host = nullptr; S.dev = nullptr; S.host_bytes = S.dev_bytes = 0; S.device = device;
    hipError_t e = hipSuccess;
    if (!S.st) e = hipStreamCreateWithFlags(&S.st, hipStreamNonBlocking);
    const size_t hw = host_bytes + host_bytes / 2 + 4096, dw = dev_bytes + dev_bytes / 2 + 4096;
    if (e == hipSuccess) e = hipHostMalloc((void**)&S.host, hw, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void**)&S.dev, dw);
    if (e == hipSuccess) { S.host_bytes = hw; S.dev_bytes = dw; }
    return e;
}
inline hipError_t slamit_scratch_reserve(SlamitScratch& S, int device, size_t bytes) { return slamit_scratch_reserve(S, device, bytes, bytes); }

#endif
