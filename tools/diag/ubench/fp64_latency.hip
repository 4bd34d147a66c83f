// Dependent-chain latency of fp64 VALU operations on gfx950, one wave per CU: cycles per operation (s_memtime).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND>
__global__ void k(double* out, unsigned long long* cyc, int iters, double seed) {
    double x = seed + threadIdx.x * 1e-9, y = 1.0000001, z = 0.999999;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) x = __builtin_fma(x, y, z);
            else if (KIND == 1) x = x * y;
            else if (KIND == 2) x = __builtin_amdgcn_rcp(x);
            else if (KIND == 3) { double r = __builtin_amdgcn_rcp(x); x = r * (2.0 - x * r); }
            else if (KIND == 4) x = x + y;
            else if (KIND == 5) { float f = (float)x; f = __builtin_fmaf(f, 1.0000001f, 0.5f); x = f; }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int KIND>
void run(const char* name, int per) {
    double* out; unsigned long long* cyc; hipMalloc(&out, 8 * 64); hipMalloc(&cyc, 8);
    const int iters = 2000;
    k<KIND><<<1, 64>>>(out, cyc, 10, 1.25);
    k<KIND><<<1, 64>>>(out, cyc, iters, 1.25);
    hipDeviceSynchronize();
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s %.1f cycles per step (%d dependent instruction(s) per step)\n", name, (double)c / (16.0 * iters), per);
}
int main() {
    run<0>("v_fma_f64 chain", 1); run<1>("v_mul_f64 chain", 1); run<4>("v_add_f64 chain", 1);
    run<2>("v_rcp_f64 chain", 1); run<3>("rcp + Newton (rcp, fma, mul)", 3); run<5>("cvt f64->f32, fma f32, cvt back", 3);
    return 0;
}
