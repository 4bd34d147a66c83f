// frame.hip — what Frame's constructor does right after ORBextractor::operator(): Frame::UndistortKeyPoints
// (ORB_SLAM2/src/Frame.cc:529-559) and Frame::AssignFeaturesToGrid (:336-357, PosInGrid :505-517), fused, on the
// keypoints where the extractor left them (include/slamit.h: slamit_frame_finish*, slamit_undistort_points).
//
// cv::undistortPoints(src, dst, K, D, Mat(), K) is restated from the published cvUndistortPoints (OpenCV 2.4): normalise,
// five fixed-point iterations of the inverse Brown model in double, re-project with P = K.  Every operation is an
// IEEE double add / mul / div in the reference's order (this file is compiled with -ffp-contract=off), so the device
// reproduces the CPU restatement bit for bit.  The grid is returned as CSR over the 64 x 48 cells in mGrid[x][y] order
// (cell = x * 48 + y), each cell's indices in keypoint order — the order Frame::GetFeaturesInArea later scans.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/slamit.h"
#include "slamit_internal.h"

#define FG_COLS 64
#define FG_ROWS 48
#define FG_CELLS (FG_COLS * FG_ROWS)
#define FF_THREADS 256

struct CamD { double fx, fy, cx, cy, k1, k2, p1, p2, k3; int identity; };

__device__ __forceinline__ void undistort_point(const CamD& c, float xin, float yin, float* xo, float* yo) {
    const double ifx = 1. / c.fx, ify = 1. / c.fy;
    double x = xin, y = yin;
    const double x0 = x = (x - c.cx) * ifx, y0 = y = (y - c.cy) * ify;
#pragma unroll 1
    for (int j = 0; j < 5; ++j) {
        const double r2 = x * x + y * y;
        // k[5..7] = 0: the numerator polynomial is 1 + ((0*r2 + 0)*r2 + 0)*r2, kept so that signed zeros round the same way
        const double icdist = (1 + ((0. * r2 + 0.) * r2 + 0.) * r2) / (1 + ((c.k3 * r2 + c.k2) * r2 + c.k1) * r2);
        const double deltaX = 2 * c.p1 * x * y + c.p2 * (r2 + 2 * x * x);
        const double deltaY = c.p1 * (r2 + 2 * y * y) + 2 * c.p2 * x * y;
        x = (x0 - deltaX) * icdist;
        y = (y0 - deltaY) * icdist;
    }
    const double xx = c.fx * x + 0. * y + c.cx, yy = 0. * x + c.fy * y + c.cy, ww = 1. / (0. * x + 0. * y + 1.);
    *xo = (float)(xx * ww);
    *yo = (float)(yy * ww);
}

__global__ __launch_bounds__(256) void undistort_kernel(CamD cam, const float* __restrict__ in, int n, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float x, y;
    undistort_point(cam, in[2 * i], in[2 * i + 1], &x, &y);
    out[2 * i] = x; out[2 * i + 1] = y;
}

// One workgroup per frame.  (1) every keypoint: undistort, cell = PosInGrid, count per cell (LDS atomics);
// (2) exclusive scan of the 3072 counts; (3) wavefront 0 places the indices IN KEYPOINT ORDER: 64 keypoints at a
// time, a lane's slot = cell start + what earlier chunks put there + the lower lanes of this chunk in the same cell.
__global__ __launch_bounds__(FF_THREADS) void frame_finish_kernel(
    CamD cam, const slamit_kp* __restrict__ kps, const int* __restrict__ n_arr, int n_fixed, int cap, float min_x, float min_y,
    float inv_w, float inv_h, slamit_kp* __restrict__ kps_un, int* __restrict__ cell_start, int* __restrict__ cell_items) {
    __shared__ int s_cnt[FG_CELLS + 1];
    __shared__ int s_wave[FF_THREADS / 64];
    extern __shared__ short s_cell[];   // cap entries
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int n = min(n_arr ? n_arr[f] : n_fixed, cap);
    const slamit_kp* K = kps + (size_t)f * cap;
    slamit_kp* U = kps_un + (size_t)f * cap;
    int* CS = cell_start + (size_t)f * (FG_CELLS + 1);
    int* CI = cell_items + (size_t)f * cap;
    for (int c = tid; c <= FG_CELLS; c += FF_THREADS) s_cnt[c] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += FF_THREADS) {
        slamit_kp k = K[i];
        if (!cam.identity) undistort_point(cam, k.x, k.y, &k.x, &k.y);
        U[i] = k;
        // Frame::PosInGrid (round = half away from zero)
        const int px = (int)roundf((k.x - min_x) * inv_w), py = (int)roundf((k.y - min_y) * inv_h);
        const bool in = !(px < 0 || px >= FG_COLS || py < 0 || py >= FG_ROWS);
        const int cell = in ? px * FG_ROWS + py : -1;
        s_cell[i] = (short)cell;
        if (in) atomicAdd(&s_cnt[cell], 1);
    }
    __syncthreads();
    // exclusive scan of s_cnt[0..FG_CELLS): 12 cells per thread
    {
        constexpr int PER = FG_CELLS / FF_THREADS;
        int loc[PER], sum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { loc[j] = s_cnt[tid * PER + j]; sum += loc[j]; }
        int incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d, 64); if (lane >= d) incl += t; }
        if (lane == 63) s_wave[tid >> 6] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += s_wave[w];
        int run = base + incl - sum;
#pragma unroll
        for (int j = 0; j < PER; ++j) { s_cnt[tid * PER + j] = run; run += loc[j]; }
        if (tid == FF_THREADS - 1) s_cnt[FG_CELLS] = run;
        __syncthreads();
    }
    for (int c = tid; c <= FG_CELLS; c += FF_THREADS) CS[c] = s_cnt[c];
    __syncthreads();
    if (tid < 64) {   // s_cnt[c] now doubles as the next free slot of cell c
        for (int i0 = 0; i0 < n; i0 += 64) {
            const int i = i0 + lane;
            const int cell = i < n ? (int)s_cell[i] : -1;
            int below = 0, total = 0;   // lanes of this chunk in my cell: below me / all
#pragma unroll 8
            for (int l = 0; l < 64; ++l) {
                const int oc = __builtin_amdgcn_readlane(cell, l);
                const int same = (oc == cell) ? 1 : 0;
                total += same;
                below += (l < lane) ? same : 0;
            }
            int slot = 0;
            if (cell >= 0) slot = s_cnt[cell] + below;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (cell >= 0) {
                CI[slot] = i;
                if (below == total - 1) s_cnt[cell] += total;   // the last lane of the cell in this chunk advances it
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

static CamD make_cam(const slamit_camera* c) {
    CamD d;
    d.fx = c->fx; d.fy = c->fy; d.cx = c->cx; d.cy = c->cy;
    d.k1 = c->k1; d.k2 = c->k2; d.p1 = c->p1; d.p2 = c->p2; d.k3 = c->k3;
    d.identity = c->k1 == 0.0f ? 1 : 0;   // Frame.cc:531: mvKeysUn = mvKeys when k1 == 0
    return d;
}

extern "C" int slamit_undistort_points(int device, const slamit_camera* cam, const float* xy_in, int n, float* xy_out) {
    if (!cam || n < 0 || (n && (!xy_in || !xy_out))) return slamit_fail(SLAMIT_ERR_ARG, "slamit_undistort_points: bad argument");
    if (n == 0) return SLAMIT_OK;
    SLAMIT_USE_DEVICE(device);
    // one pinned staging block + one device slab per host thread (slamit_internal.h): [in | out], one copy each way
    const size_t half = (8 * (size_t)n + 255) & ~(size_t)255;
    static thread_local SlamitScratch S;
    hipError_t e = slamit_scratch_reserve(S, device, 2 * half);
    if (e == hipSuccess) {
        memcpy(S.host, xy_in, 8 * (size_t)n);
        e = hipMemcpyAsync(S.dev, S.host, 8 * (size_t)n, hipMemcpyHostToDevice, S.st);
    }
    if (e == hipSuccess) {
        CamD c = make_cam(cam);
        c.identity = 0;   // cv::undistortPoints itself has no shortcut
        hipLaunchKernelGGL(undistort_kernel, dim3((n + 255) / 256), dim3(256), 0, S.st, c, reinterpret_cast<const float*>(S.dev), n, reinterpret_cast<float*>(S.dev + half));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(S.host + half, S.dev + half, 8 * (size_t)n, hipMemcpyDeviceToHost, S.st);
    if (e == hipSuccess) e = hipStreamSynchronize(S.st);
    if (e == hipSuccess) memcpy(xy_out, S.host + half, 8 * (size_t)n);
    if (e != hipSuccess) return slamit_fail_hip(e, "slamit_undistort_points");
    return SLAMIT_OK;
}

extern "C" int slamit_frame_finish_batch_dev(int device, const slamit_camera* cam, const slamit_kp* d_kps, const int32_t* d_n, int cap,
                                             int nframes, float min_x, float min_y, float inv_w, float inv_h, slamit_kp* d_kps_un,
                                             int32_t* d_cell_start, int32_t* d_cell_items, void* stream) {
    if (!cam || !d_kps || !d_n || !d_kps_un || !d_cell_start || !d_cell_items || cap < 0 || nframes < 0)
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_frame_finish_batch_dev: bad argument");
    if (cap > SLAMIT_FRAME_MAX_KP) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_frame_finish_batch_dev: cap > SLAMIT_FRAME_MAX_KP");
    if (nframes == 0) return SLAMIT_OK;
    SLAMIT_USE_DEVICE(device);
    hipLaunchKernelGGL(frame_finish_kernel, dim3(nframes), dim3(FF_THREADS), sizeof(short) * (size_t)std::max(cap, 1), (hipStream_t)stream,
                       make_cam(cam), d_kps, d_n, 0, cap, min_x, min_y, inv_w, inv_h, d_kps_un, d_cell_start, d_cell_items);
    HIP_TRY(hipGetLastError());
    return SLAMIT_OK;
}

extern "C" int slamit_frame_finish(int device, const slamit_camera* cam, const slamit_kp* kps, int n, float min_x, float min_y,
                                   float inv_w, float inv_h, slamit_kp* kps_un, int32_t* cell_start, int32_t* cell_items) {
    if (!cam || n < 0 || !cell_start || (n && (!kps || !kps_un || !cell_items)))
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_frame_finish: bad argument");
    if (n > SLAMIT_FRAME_MAX_KP) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_frame_finish: more than SLAMIT_FRAME_MAX_KP keypoints");
    SLAMIT_USE_DEVICE(device);
    const int cap = std::max(n, 1);
    // one pinned staging block + one device slab per host thread: [keypoints in | keypoints out | cell_start | cell_items]
    const size_t kb = (sizeof(slamit_kp) * (size_t)cap + 255) & ~(size_t)255, cs = (sizeof(int) * (FG_CELLS + 1) + 255) & ~(size_t)255;
    const size_t o_un = kb, o_cs = 2 * kb, o_ci = o_cs + cs, bytes = o_ci + sizeof(int) * (size_t)cap;
    static thread_local SlamitScratch S;
    hipError_t e = slamit_scratch_reserve(S, device, bytes);
    if (e == hipSuccess && n) {
        memcpy(S.host, kps, sizeof(slamit_kp) * (size_t)n);
        e = hipMemcpyAsync(S.dev, S.host, sizeof(slamit_kp) * (size_t)n, hipMemcpyHostToDevice, S.st);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(frame_finish_kernel, dim3(1), dim3(FF_THREADS), sizeof(short) * (size_t)cap, S.st, make_cam(cam),
                           reinterpret_cast<const slamit_kp*>(S.dev), (const int*)nullptr, n, cap, min_x, min_y, inv_w, inv_h,
                           reinterpret_cast<slamit_kp*>(S.dev + o_un), reinterpret_cast<int*>(S.dev + o_cs), reinterpret_cast<int*>(S.dev + o_ci));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(S.host + o_un, S.dev + o_un, bytes - o_un, hipMemcpyDeviceToHost, S.st);
    if (e == hipSuccess) e = hipStreamSynchronize(S.st);
    if (e == hipSuccess) {
        if (n) memcpy(kps_un, S.host + o_un, sizeof(slamit_kp) * (size_t)n);
        memcpy(cell_start, S.host + o_cs, sizeof(int) * (FG_CELLS + 1));
        if (n) memcpy(cell_items, S.host + o_ci, sizeof(int) * (size_t)std::min(n, cell_start[FG_CELLS]));
    }
    if (e != hipSuccess) return slamit_fail_hip(e, "slamit_frame_finish");
    return SLAMIT_OK;
}
