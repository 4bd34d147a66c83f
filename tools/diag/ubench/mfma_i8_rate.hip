// Issue rate of the int8 MFMAs on gfx950: ns and shader cycles per instruction per SIMD, one or two waves per SIMD,
// independent accumulators vs one dependent chain.  hipcc -O3 --offload-arch=gfx950 mfma_i8_rate.hip -o mfma_i8_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v4acc __attribute__((ext_vector_type(4)));

template <int KIND, int CHAINS>
__global__ __launch_bounds__(256) void k(int iters, int* out, unsigned long long* cyc) {
    v4i a = {(int)threadIdx.x, 2, 3, 4}, b = {5, 6, (int)blockIdx.x, 8};
    v16i c[4] = {};
    v4acc d[4] = {};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) c[u % CHAINS] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c[u % CHAINS], 0, 0, 0);
            else d[u % CHAINS] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, d[u % CHAINS], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int u = 0; u < 4; ++u) { for (int r = 0; r < 16; ++r) s += c[u][r]; for (int r = 0; r < 4; ++r) s += d[u][r]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// the matcher's inner loop: the A operand of each k-step is built by VALU (shift + bitop3 per dword) right before the 4 MFMAs that use it
template <int MODE>
__global__ __launch_bounds__(256) void kmix(int iters, int* out, unsigned long long* cyc, const uint4* src) {
    v4i b[4][8];
    for (int u = 0; u < 4; ++u) for (int s = 0; s < 8; ++s) b[u][s] = v4i{(int)threadIdx.x + u, s, 3, (int)blockIdx.x};
    v16i c[4] = {};
    uint4 w = src[threadIdx.x];
    const unsigned hi = 0x80808080u, mid = 0x40404040u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 2) { w.x += i; w.y ^= i; }
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            v4i a;
            if (MODE == 0) { a.x = w.x; a.y = w.y; a.z = w.z; a.w = w.w; }   // operand straight from registers
            else {
                a.x = (int)__builtin_amdgcn_bitop3_b32(w.x << (7 - s), hi, mid, 0xAE); a.y = (int)__builtin_amdgcn_bitop3_b32(w.y << (7 - s), hi, mid, 0xAE);
                a.z = (int)__builtin_amdgcn_bitop3_b32(w.z << (7 - s), hi, mid, 0xAE); a.w = (int)__builtin_amdgcn_bitop3_b32(w.w << (7 - s), hi, mid, 0xAE);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[u][s], c[u], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int sum = 0;
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) sum += c[u][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void runmix(const char* name, int blocks) {
    int* out; unsigned long long* cyc; uint4* src;
    hipMalloc(&out, sizeof(int) * blocks * 256); hipMalloc(&cyc, 8); hipMalloc(&src, 16 * 256); hipMemset(src, 0x5a, 16 * 256);
    const int iters = 500;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kmix<MODE><<<blocks, 256>>>(10, out, cyc, src);
    hipEventRecord(e0);
    kmix<MODE><<<blocks, 256>>>(iters, out, cyc, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 32.0 * iters * (blocks / 256.0);
    printf("%-44s blocks %4d: %.2f ns per MFMA per SIMD, %.1f ticks per MFMA of one wave\n", name, blocks, ms * 1e6 / n, (double)hc / (32.0 * iters));
    hipFree(out); hipFree(cyc); hipFree(src);
}

// the matcher's epilogue alone: 4 x 16 keys folded into (best, second) with v_min_i32 + v_med3_i32 (asm) or min / max / min
__device__ __forceinline__ int med3_asm(int a, int b, int c, int after) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(after)); return r; }
template <int MODE>
__global__ __launch_bounds__(256) void kepi(int iters, int* out, unsigned long long* cyc, const int* src) {
    int c[4][16];
    for (int u = 0; u < 4; ++u) for (int r = 0; r < 16; ++r) c[u][r] = src[(threadIdx.x + 17 * u + 3 * r) & 255];
    int kb[4] = {1 << 30, 1 << 30, 1 << 30, 1 << 30}, ks[4] = {1 << 30, 1 << 30, 1 << 30, 1 << 30};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = c[u][r] + i;
                const int n = min(kb[u], k);
                if (MODE == 0) ks[u] = med3_asm(kb[u], ks[u], k, n);
                else ks[u] = max(kb[u], min(ks[u], k));
                kb[u] = n;
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = kb[0] + kb[1] + kb[2] + kb[3] + ks[0] + ks[1] + ks[2] + ks[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void runepi(const char* name, int blocks) {
    int* out; unsigned long long* cyc; int* src;
    hipMalloc(&out, sizeof(int) * blocks * 256); hipMalloc(&cyc, 8); hipMalloc(&src, 1024); hipMemset(src, 0x11, 1024);
    const int iters = 1000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kepi<MODE><<<blocks, 256>>>(10, out, cyc, src);
    hipEventRecord(e0);
    kepi<MODE><<<blocks, 256>>>(iters, out, cyc, src);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s blocks %4d: %.1f ns per 64-key epilogue per SIMD, %.0f ticks per epilogue of one wave\n", name, blocks, ms * 1e6 / (iters * (blocks / 256.0)), (double)hc / iters);
    hipFree(out); hipFree(cyc); hipFree(src);
}

template <int KIND, int CHAINS>
void run(const char* name, int blocks) {
    int* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(int) * blocks * 256); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, CHAINS><<<blocks, 256>>>(10, out, cyc);
    hipEventRecord(e0);
    k<KIND, CHAINS><<<blocks, 256>>>(iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 8.0 * iters * (blocks / 256.0);   // MFMAs per SIMD (one wave of a block per SIMD)
    printf("%-28s blocks %4d (%.0f waves/SIMD): %.2f ns per MFMA per SIMD, %.1f memtime ticks per MFMA of one wave\n", name, blocks, blocks / 256.0,
           ms * 1e6 / n, (double)hc / (8.0 * iters));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0, 4>("32x32x32 i8, 4 chains", 256);
    run<0, 4>("32x32x32 i8, 4 chains", 512);
    run<0, 1>("32x32x32 i8, 1 chain", 256);
    run<0, 2>("32x32x32 i8, 2 chains", 256);
    run<1, 4>("16x16x64 i8, 4 chains", 256);
    run<1, 4>("16x16x64 i8, 4 chains", 512);
    run<1, 1>("16x16x64 i8, 1 chain", 256);
    runepi<0>("epilogue, add + min + med3(asm) per key", 256);
    runepi<0>("epilogue, add + min + med3(asm) per key", 512);
    runepi<1>("epilogue, add + min + max + min per key", 256);
    runepi<1>("epilogue, add + min + max + min per key", 512);
    runmix<0>("loop of 8 steps x 4 MFMA, A from registers", 256);
    runmix<0>("loop of 8 steps x 4 MFMA, A from registers", 512);
    runmix<1>("... A unpacked by VALU per step (invariant w)", 256);
    runmix<2>("... A unpacked by VALU per step (w changes)", 256);
    runmix<2>("... A unpacked by VALU per step (w changes)", 512);
    return 0;
}
