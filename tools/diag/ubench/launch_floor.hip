// Cost of a (nearly) empty launch on gfx950 as a function of grid size and VGPR allocation: per-launch time of 200
// back-to-back launches on one stream (HIP events).  hipcc -O3 --offload-arch=gfx950 launch_floor.hip -o launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int BIGREG>
__global__ __launch_bounds__(256) void k(int* out, int n) {
    if (BIGREG) asm volatile("v_mov_b32 v249, 0" ::: "v249");
    if (n < 0) out[threadIdx.x] = 1;
}
template <int BIGREG>
void run(const char* name, int blocks) {
    int* out; hipMalloc(&out, 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; ++i) k<BIGREG><<<blocks, 256>>>(out, 0);
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) k<BIGREG><<<blocks, 256>>>(out, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s blocks %5d: %.2f us per launch\n", name, blocks, ms * 1e3 / 200);
    hipFree(out);
}
int main() {
    for (int b : {1, 64, 256, 576, 1024, 4096, 16384}) { run<0>("empty, few VGPRs", b); run<1>("empty, 250 VGPRs", b); }
    return 0;
}
