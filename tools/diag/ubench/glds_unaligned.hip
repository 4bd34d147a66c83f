// Does an LDS-DMA load (global_load_lds_dword / _dwordx4) accept a global source address that is not dword aligned?
// The FAST cell window starts at an arbitrary byte of an image row; if the DMA takes it, a cell's tile can be staged by
// two wave instructions with no VGPR traffic.  Prints per (size, shift) whether LDS holds the bytes at src + shift.
// hipcc -O3 --offload-arch=gfx950 glds_unaligned.hip -o glds_unaligned
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int SIZE>
__global__ __launch_bounds__(64) void k(const uint8_t* src, int shift, int stride, uint8_t* out) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[64 * 16];
    const int lane = threadIdx.x;
    // lane -> (row, chunk): three 16-byte chunks (or twelve dwords) per 48-byte tile row, rows `stride` apart in memory
    const int per_row = 48 / SIZE;
    const int row = lane / per_row, c = lane - row * per_row;
    const uint8_t* g = src + shift + row * stride + c * SIZE;
    if constexpr (SIZE == 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)tile, 4, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)tile, 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < 64 * SIZE; i += 64) out[i] = tile[i];
}

int main() {
    const int stride = 640, n = 64 * stride;
    uint8_t* h = (uint8_t*)malloc(n);
    for (int i = 0; i < n; ++i) h[i] = (uint8_t)((i * 131) ^ (i >> 8));
    uint8_t *d, *o;
    (void)hipMalloc(&d, n); (void)hipMalloc(&o, 1024);
    (void)hipMemcpy(d, h, n, hipMemcpyHostToDevice);
    uint8_t got[1024];
    int bad_total = 0;
    for (int size : {4, 16}) {
        for (int shift = 0; shift < 8; ++shift) {
            (void)hipMemset(o, 0xEE, 1024);
            if (size == 4) k<4><<<1, 64>>>(d, shift, stride, o); else k<16><<<1, 64>>>(d, shift, stride, o);
            hipError_t e = hipDeviceSynchronize();
            if (e != hipSuccess) { printf("size %d shift %d: %s\n", size, shift, hipGetErrorString(e)); return 1; }
            (void)hipMemcpy(got, o, 1024, hipMemcpyDeviceToHost);
            int bad = 0;
            const int per_row = 48 / size;
            for (int lane = 0; lane < 64; ++lane)
                for (int b = 0; b < size; ++b) {
                    const int row = lane / per_row, c = lane - row * per_row;
                    if (got[lane * size + b] != h[shift + row * stride + c * size + b]) ++bad;
                }
            printf("size %2d shift %d: %s (%d wrong bytes)\n", size, shift, bad ? "MISMATCH" : "ok", bad);
            bad_total += bad;
        }
    }
    printf(bad_total ? "LDS-DMA needs an aligned source\n" : "LDS-DMA takes unaligned global sources\n");
    return 0;
}
