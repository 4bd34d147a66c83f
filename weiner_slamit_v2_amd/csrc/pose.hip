// pose.hip — Optimizer::PoseOptimization on the GPU (include/slamit.h, slamit_pose_*).
//
// Reference: ORB_SLAM2/src/Optimizer.cc:239-451 driving g2o (BlockSolver_6_3 + LinearSolverDense +
// Levenberg) over EdgeSE3ProjectXYZOnlyPose edges (Thirdparty/g2o/g2o/types/types_six_dof_expmap.{h:143-170,
// cpp:266-288}) and, for the keypoints of a stereo / RGB-D frame that have a right-image column, EdgeStereoSE3ProjectXYZOnlyPose
// edges ({h:174-202, cpp:299-306, 335-364}; Optimizer.cc:319-356: three residual rows, Huber width sqrt(7.815), gate 7.815f).  The system is a single 6x6 block, so the whole schedule — 4 rounds x <= 10 LM
// iterations x <= 10 trials, the (float)chi2 > 5.991f relabelling between rounds, the kernel drop after
// the third round — runs inside ONE workgroup per frame with no host round trip; a batch of frames is
// one launch.  Reductions are fixed-order (lane-strided partial sums, shuffle tree, 4 waves in order).
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/slamit.h"
#include "se3_device.h"
#include "slamit_internal.h"

struct PoseFrame {
    int32_t n;
    const double* pose_in;   // 12
    const double* intr;      // 4
    const double* xw;        // n x 3
    const double* uv;        // n x 2
    const double* w;         // n
    double* chi2;            // n scratch
    uint8_t* outlier;        // n out
    double* pose_out;        // 12
    int32_t* n_inliers;      // 1
    int32_t* n_its;          // 4
    double* chi2_round;      // 4
    const double* ur;        // n: right-image column, < 0 on a monocular correspondence; null when the frame has none (Optimizer.cc:281)
    double bf;               // Frame::mbf
};
// The same record as the kernel sees it: pointers in the global address space.  (Read out of a struct in memory a plain
// pointer is generic, and every access through it a flat_load; host code that fills PoseFrame is parsed in the device
// pass too, so the qualified twin is a separate type.)
#if defined(__HIP_DEVICE_COMPILE__)
struct PoseFrameG {
    int32_t n;
    const __attribute__((address_space(1))) double* pose_in;   // 12
    const __attribute__((address_space(1))) double* intr;      // 4
    const __attribute__((address_space(1))) double* xw;        // n x 3
    const __attribute__((address_space(1))) double* uv;        // n x 2
    const __attribute__((address_space(1))) double* w;         // n
    __attribute__((address_space(1))) double* chi2;            // n scratch
    __attribute__((address_space(1))) uint8_t* outlier;        // n out
    __attribute__((address_space(1))) double* pose_out;        // 12
    __attribute__((address_space(1))) int32_t* n_inliers;      // 1
    __attribute__((address_space(1))) int32_t* n_its;          // 4
    __attribute__((address_space(1))) double* chi2_round;      // 4
    const __attribute__((address_space(1))) double* ur;        // n or null
    double bf;
};
static_assert(sizeof(PoseFrameG) == sizeof(PoseFrame), "PoseFrameG mirrors PoseFrame");
#else
typedef PoseFrame PoseFrameG;   // the host pass only needs the name
#endif

namespace {

__device__ __forceinline__ double block_sum(double v, double* sh) {  // 256 threads, result in every thread
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

#define POSE_DELTA_STEREO ((double)(float)sqrt(7.815))   // deltaStereo, Optimizer.cc:265

__device__ __forceinline__ bool pose_is_stereo(const PoseFrameG& F, int e) { return F.ur && !(F.ur[e] < 0.0); }

// error of a stereo edge at camera-frame point Xc: EdgeStereoSE3ProjectXYZOnlyPose::cam_project (types_six_dof_expmap.cpp:299-306),
// `const float invz = 1.0f / z` (the quotient rounded to float), bf a double member.  Returns chi2.
__device__ __forceinline__ double pose_stereo_error(const PoseFrameG& F, int e, const double* Xc, double* err /*[3]*/) {
    const float invz = (float)(1.0 / Xc[2]);
    const double r0 = Xc[0] * (double)invz * F.intr[0] + F.intr[2];
    const double r1 = Xc[1] * (double)invz * F.intr[1] + F.intr[3];
    const double r2 = r0 - F.bf * (double)invz;
    err[0] = F.uv[2 * e] - r0; err[1] = F.uv[2 * e + 1] - r1; err[2] = F.ur[e] - r2;
    const double w = F.w[e];
    return err[0] * w * err[0] + err[1] * w * err[1] + err[2] * w * err[2];
}

// residual + chi2 of every active edge at pose T; returns the robust cost
__device__ double pose_errors(const PoseFrameG& F, const double* T, const uint8_t* active, int robust, double delta, double* sh) {
    const double dsqr = (double)(float)(delta * delta);   // RobustKernelHuber::dsqr is a float member (g2o/core/robust_kernel_impl.h:84)
    const double delta_s = POSE_DELTA_STEREO, dsqr_s = (double)(float)(delta_s * delta_s);
    const double fx = F.intr[0], fy = F.intr[1], cx = F.intr[2], cy = F.intr[3];
    double part = 0;
    for (int e = threadIdx.x; e < F.n; e += 256) {
        if (!active[e]) continue;
        double Xc[3];
        quat_rot(T, F.xw + 3 * e, Xc);
        Xc[0] += T[4]; Xc[1] += T[5]; Xc[2] += T[6];
        if (pose_is_stereo(F, e)) {
            double err[3];
            const double c2 = pose_stereo_error(F, e, Xc, err);
            F.chi2[e] = c2;
            part += (robust && c2 > dsqr_s) ? 2 * sqrt(c2) * delta_s - dsqr_s : c2;
            continue;
        }
        const double e0 = F.uv[2 * e] - (Xc[0] / Xc[2] * fx + cx), e1 = F.uv[2 * e + 1] - (Xc[1] / Xc[2] * fy + cy);
        const double w = F.w[e];
        const double c2 = e0 * w * e0 + e1 * w * e1;
        F.chi2[e] = c2;
        part += (robust && c2 > dsqr) ? 2 * sqrt(c2) * delta - dsqr : c2;
    }
    return block_sum(part, sh);
}

// 6x6 LDLt without pivoting (H + lambda I) x = b; false on a zero pivot
__device__ bool solve6(const double* H, double lambda, const double* b, double* x) {
    double A[36];
    for (int i = 0; i < 36; ++i) A[i] = H[i] + (i % 7 == 0 ? lambda : 0.0);
    for (int j = 0; j < 6; ++j) {
        double d = A[7 * j];
        for (int k = 0; k < j; ++k) d -= A[6 * j + k] * A[6 * j + k] * A[7 * k];
        if (d == 0.0 || !(fabs(d) <= DBL_MAX)) return false;
        A[7 * j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double s = A[6 * i + j];
            for (int k = 0; k < j; ++k) s -= A[6 * i + k] * A[6 * j + k] * A[7 * k];
            A[6 * i + j] = s / d;
        }
    }
    for (int i = 0; i < 6; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[6 * i + k] * x[k]; x[i] = s; }
    for (int i = 0; i < 6; ++i) x[i] /= A[7 * i];
    for (int i = 5; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < 6; ++k) s -= A[6 * k + i] * x[k]; x[i] = s; }
    return true;
}

}  // namespace

__global__ __launch_bounds__(256) void pose_opt_kernel(const PoseFrame* frames) {
    const PoseFrameG F = reinterpret_cast<const PoseFrameG*>(frames)[blockIdx.x];
    const int tid = threadIdx.x, n = F.n;
    __shared__ double sh[4];
    __shared__ double sT0[7], sT[7], sTbak[7], sH[36], sb[6], sx[6];
    __shared__ double s_lambda, s_ni, s_cur, s_rho;
    __shared__ int s_ok2, s_cnt;
    extern __shared__ uint8_t s_act[];  // n bytes: edge is at level 0
    if (n < 3) {  // Optimizer.cc:364-365
        if (tid < 12) F.pose_out[tid] = F.pose_in[tid];
        if (tid < n) F.outlier[tid] = 0;
        if (tid == 0) { *F.n_inliers = 0; for (int r = 0; r < 4; ++r) { F.n_its[r] = 0; F.chi2_round[r] = 0; } }
        return;
    }
    if (tid == 0) {
        double R[9], q[4];
        for (int i = 0; i < 9; ++i) R[i] = F.pose_in[i];
        R_to_quat(R, q);
        quat_normalize(q);
        for (int i = 0; i < 4; ++i) sT0[i] = q[i];
        for (int i = 0; i < 3; ++i) sT0[4 + i] = F.pose_in[9 + i];
        for (int r = 0; r < 4; ++r) { F.n_its[r] = 0; F.chi2_round[r] = 0; }
    }
    for (int e = tid; e < n; e += 256) { s_act[e] = 1; F.outlier[e] = 0; F.chi2[e] = 0; }
    __syncthreads();
    const double delta = (double)(float)sqrt(5.991);
    int robust = 1, nBad = 0;
    for (int round = 0; round < 4; ++round) {
        if (tid < 7) sT[tid] = sT0[tid];  // every round restarts from the input pose (:373)
        int nact = 0;
        for (int e = tid; e < n; e += 256) nact += s_act[e];
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        if (nact) atomicAdd(&s_cnt, nact);
        __syncthreads();
        const bool any_active = s_cnt > 0;
        int done = 0;
        double lastChi = 0;
        if (any_active) {
            int lm_nBad = 0;
            bool ok = true;
            for (int it = 0; it < 10 && ok; ++it) {
                // g2o re-evaluates the errors at the top of every iteration; after an accepted trial (the only way to get here
                // with it > 0) they are the ones that trial just computed at this very pose: same bits, one pass saved
                const double currentChi0 = it == 0 ? pose_errors(F, sT, s_act, robust, delta, sh) : s_cur;
                // ---- normal equations H (21 unique), b (6): per-thread partials, shuffle tree, 4 waves in order ----
                double h[21], bb[6];
                for (int i = 0; i < 21; ++i) h[i] = 0;
                for (int i = 0; i < 6; ++i) bb[i] = 0;
                const double dsqr = (double)(float)(delta * delta), fx = F.intr[0], fy = F.intr[1];
                for (int e = tid; e < n; e += 256) {
                    if (!s_act[e]) continue;
                    double Xc[3];
                    quat_rot(sT, F.xw + 3 * e, Xc);
                    Xc[0] += sT[4]; Xc[1] += sT[5]; Xc[2] += sT[6];
                    const double x = Xc[0], y = Xc[1], invz = 1.0 / Xc[2], invz_2 = invz * invz;
                    double J[12];
                    J[0] = x * y * invz_2 * fx; J[1] = -(1 + (x * x * invz_2)) * fx; J[2] = y * invz * fx;
                    J[3] = -invz * fx; J[4] = 0; J[5] = x * invz_2 * fx;
                    J[6] = (1 + y * y * invz_2) * fy; J[7] = -x * y * invz_2 * fy; J[8] = -x * invz * fy;
                    J[9] = 0; J[10] = -invz * fy; J[11] = y * invz_2 * fy;
                    const double w = F.w[e], c2 = F.chi2[e];
                    if (pose_is_stereo(F, e)) {   // third row: EdgeStereoSE3ProjectXYZOnlyPose::linearizeOplus (types_six_dof_expmap.cpp:335-364)
                        const double bf = F.bf, delta_s = POSE_DELTA_STEREO, dsqr_s = (double)(float)(delta_s * delta_s);
                        const double J2[6] = {J[0] - bf * y * invz_2, J[1] + bf * x * invz_2, J[2], J[3], 0.0, J[5] - bf * invz_2};
                        double err[3];
                        pose_stereo_error(F, e, Xc, err);
                        const double rho1s = (robust && c2 > dsqr_s) ? delta_s / sqrt(c2) : 1.0;
                        const double wOs = rho1s * w;
                        int k = 0;
#pragma unroll
                        for (int a = 0; a < 6; ++a) {
                            bb[a] -= rho1s * (J[a] * w * err[0] + J[6 + a] * w * err[1] + J2[a] * w * err[2]);
#pragma unroll
                            for (int c = a; c < 6; ++c) h[k++] += (J[a] * J[c] + J[6 + a] * J[6 + c] + J2[a] * J2[c]) * wOs;
                        }
                        continue;
                    }
                    const double e0 = F.uv[2 * e] - (x * invz * fx + F.intr[2]), e1 = F.uv[2 * e + 1] - (y * invz * fy + F.intr[3]);
                    const double rho1 = (robust && c2 > dsqr) ? delta / sqrt(c2) : 1.0;
                    const double wO = rho1 * w;
                    int k = 0;
#pragma unroll
                    for (int a = 0; a < 6; ++a) {
                        bb[a] -= rho1 * (J[a] * w * e0 + J[6 + a] * w * e1);
#pragma unroll
                        for (int c = a; c < 6; ++c) h[k++] += (J[a] * J[c] + J[6 + a] * J[6 + c]) * wO;
                    }
                }
                __shared__ double red[4][27];
                for (int i = 0; i < 21; ++i) { const double v = wave_sum(h[i]); if ((tid & 63) == 0) red[tid >> 6][i] = v; }
                for (int i = 0; i < 6; ++i) { const double v = wave_sum(bb[i]); if ((tid & 63) == 0) red[tid >> 6][21 + i] = v; }
                __syncthreads();
                if (tid == 0) {
                    int k = 0;
                    for (int a = 0; a < 6; ++a)
                        for (int c = a; c < 6; ++c) { const double v = red[0][k] + red[1][k] + red[2][k] + red[3][k]; sH[6 * a + c] = v; sH[6 * c + a] = v; ++k; }
                    for (int a = 0; a < 6; ++a) sb[a] = red[0][21 + a] + red[1][21 + a] + red[2][21 + a] + red[3][21 + a];
                    if (it == 0) {
                        double m = 0;
                        for (int j = 0; j < 6; ++j) m = fmax(m, fabs(sH[7 * j]));
                        s_lambda = 1e-5 * m; s_ni = 2;
                    }
                    s_cur = currentChi0;
                }
                if (it == 0) lm_nBad = 0;
                __syncthreads();
                const double iniChi = currentChi0;
                int qmax = 0;
                double rho = 0, tempChi = currentChi0;
                do {
                    if (tid == 0) {
                        for (int i = 0; i < 7; ++i) sTbak[i] = sT[i];
                        double x[6];
                        const bool ok2 = solve6(sH, s_lambda, sb, x);
                        if (ok2) { double T[7]; for (int i = 0; i < 7; ++i) T[i] = sT[i]; pose_oplus(T, x); for (int i = 0; i < 7; ++i) sT[i] = T[i]; }
                        else for (int i = 0; i < 6; ++i) x[i] = 0;
                        for (int i = 0; i < 6; ++i) sx[i] = x[i];
                        s_ok2 = ok2;
                    }
                    __syncthreads();
                    tempChi = pose_errors(F, sT, s_act, robust, delta, sh);
                    if (!s_ok2) tempChi = DBL_MAX;
                    if (tid == 0) {
                        double scale = 0;
                        for (int k = 0; k < 6; ++k) scale += sx[k] * (s_lambda * sx[k] + sb[k]);
                        double r = (s_cur - tempChi) / (scale + 1e-3);
                        if (r > 0 && fabs(tempChi) <= DBL_MAX) {
                            double alpha = fmin(1. - pow((2 * r - 1), 3), 2. / 3.);
                            s_lambda *= fmax(1. / 3., alpha);
                            s_ni = 2; s_cur = tempChi;
                        } else {
                            s_lambda *= s_ni; s_ni *= 2;
                            for (int i = 0; i < 7; ++i) sT[i] = sTbak[i];
                        }
                        s_rho = r;
                    }
                    __syncthreads();
                    rho = s_rho;
                    ++qmax;
                } while (rho < 0 && qmax < 10);
                ++done;
                lastChi = tempChi;
                if (qmax == 10 || rho == 0) { ok = false; continue; }
                if ((iniChi - s_cur) * 1e3 < iniChi) ++lm_nBad; else lm_nBad = 0;
                if (lm_nBad >= 3) ok = false;
            }
        }
        // ---- relabel (:374-404): outliers are re-evaluated at the new pose, (float)chi2 vs 5.991f ----
        const double fx = F.intr[0], fy = F.intr[1], cx = F.intr[2], cy = F.intr[3];
        int bad = 0;
        for (int e = tid; e < n; e += 256) {
            double c2 = F.chi2[e];
            if (F.outlier[e]) {
                double Xc[3];
                quat_rot(sT, F.xw + 3 * e, Xc);
                Xc[0] += sT[4]; Xc[1] += sT[5]; Xc[2] += sT[6];
                if (pose_is_stereo(F, e)) {
                    double err[3];
                    c2 = pose_stereo_error(F, e, Xc, err);
                } else {
                const double e0 = F.uv[2 * e] - (Xc[0] / Xc[2] * fx + cx), e1 = F.uv[2 * e + 1] - (Xc[1] / Xc[2] * fy + cy);
                const double w = F.w[e];
                c2 = e0 * w * e0 + e1 * w * e1;
                }
                F.chi2[e] = c2;
            }
            const bool out = (float)c2 > (pose_is_stereo(F, e) ? 7.815f : 5.991f);   // chi2Mono / chi2Stereo, Optimizer.cc:369-370
            F.outlier[e] = out;
            s_act[e] = !out;
            bad += out;
        }
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        if (bad) atomicAdd(&s_cnt, bad);
        __syncthreads();
        nBad = s_cnt;
        if (round == 2) robust = 0;
        if (tid == 0) { F.n_its[round] = done; F.chi2_round[round] = lastChi; }
        __syncthreads();
        if (n < 10) break;  // optimizer.edges().size() < 10
    }
    if (tid == 0) {
        double R[9];
        quat_to_R(sT, R);
        for (int i = 0; i < 9; ++i) F.pose_out[i] = R[i];
        for (int i = 0; i < 3; ++i) F.pose_out[9 + i] = sT[4 + i];
        *F.n_inliers = n - nBad;
    }
}

extern "C" {

int slamit_pose_optimize_batch(int device, int nframes, const slamit_pose_problem* probs, slamit_pose_result* results) {
    if (nframes < 0 || (nframes && (!probs || !results))) return slamit_fail(SLAMIT_ERR_ARG, "slamit_pose_optimize_batch: bad argument");
    if (nframes == 0) return SLAMIT_OK;
    SLAMIT_USE_DEVICE(device);
    // one slab per host thread (slamit_internal.h): [doubles of every frame | ints | PoseFrame records | flags], one copy each way
    size_t total = 0, flag_total = 0;
    int nmax = 1;
    std::vector<size_t> off(nframes), foff(nframes);
    for (int f = 0; f < nframes; ++f) {
        const slamit_pose_problem& P = probs[f];
        if (P.n < 0 || !P.pose || !P.intr || (P.n && (!P.xw || !P.uv || !P.inv_sigma2)) || !results[f].pose || (P.n && !results[f].outlier))
            return slamit_fail(SLAMIT_ERR_ARG, "slamit_pose_optimize_batch: null array");
        off[f] = total; foff[f] = flag_total;
        // per frame: pose 12 | intr 4 | xw 3n | uv 2n | w n | chi2 n | pose_out 12 | chi2_round 4 [| ur n]   (doubles)
        total += 32 + (size_t)(P.ur ? 8 : 7) * P.n;
        flag_total += ((size_t)P.n + 15) & ~(size_t)7;
        nmax = std::max(nmax, (int)P.n);
    }
    const size_t o_ints = sizeof(double) * total, o_frames = (o_ints + sizeof(int32_t) * 5 * nframes + 15) & ~(size_t)15;
    const size_t o_flags = o_frames + sizeof(PoseFrame) * nframes, bytes = o_flags + flag_total;
    static thread_local SlamitScratch S;
    hipError_t e = slamit_scratch_reserve(S, device, bytes);
    if (e != hipSuccess) return slamit_fail_hip(e, "slamit_pose_optimize_batch");
    double* stage = reinterpret_cast<double*>(S.host);
    double* d_buf = reinterpret_cast<double*>(S.dev);
    int32_t* d_ints = reinterpret_cast<int32_t*>(S.dev + o_ints);
    PoseFrame* fr = reinterpret_cast<PoseFrame*>(S.host + o_frames);
    uint8_t* d_flags = S.dev + o_flags;
    for (int f = 0; f < nframes; ++f) {
        const slamit_pose_problem& P = probs[f];
        double* h = stage + off[f];
        double* d = d_buf + off[f];
        const size_t n = P.n;
        memcpy(h, P.pose, 96); memcpy(h + 12, P.intr, 32);
        if (n) { memcpy(h + 16, P.xw, 24 * n); memcpy(h + 16 + 3 * n, P.uv, 16 * n); memcpy(h + 16 + 5 * n, P.inv_sigma2, 8 * n); }
        PoseFrame& F = fr[f];
        F.n = P.n; F.pose_in = d; F.intr = d + 12; F.xw = d + 16; F.uv = d + 16 + 3 * n; F.w = d + 16 + 5 * n;
        F.chi2 = d + 16 + 6 * n; F.pose_out = d + 16 + 7 * n; F.chi2_round = d + 28 + 7 * n;
        F.ur = nullptr; F.bf = 0.0;
        if (P.ur && n) { memcpy(h + 32 + 7 * n, P.ur, 8 * n); F.ur = d + 32 + 7 * n; F.bf = P.bf; }
        F.outlier = d_flags + foff[f];
        F.n_inliers = d_ints + 5 * f; F.n_its = d_ints + 5 * f + 1;
    }
    e = hipMemcpyAsync(S.dev, S.host, o_flags, hipMemcpyHostToDevice, S.st);
    if (e == hipSuccess) {
        if (nmax > 48 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void*>(pose_opt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, nmax + 16);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(pose_opt_kernel, dim3(nframes), dim3(256), (size_t)nmax + 16, S.st, reinterpret_cast<const PoseFrame*>(S.dev + o_frames));
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(S.host, S.dev, bytes, hipMemcpyDeviceToHost, S.st);
    if (e == hipSuccess) e = hipStreamSynchronize(S.st);
    if (e != hipSuccess) return slamit_fail_hip(e, "slamit_pose_optimize_batch");
    const int32_t* ints = reinterpret_cast<const int32_t*>(S.host + o_ints);
    for (int f = 0; f < nframes; ++f) {
        const size_t n = probs[f].n;
        const double* h = stage + off[f];
        memcpy(results[f].pose, h + 16 + 7 * n, 96);
        for (int r = 0; r < 4; ++r) { results[f].chi2[r] = h[28 + 7 * n + r]; results[f].n_its[r] = ints[5 * f + 1 + r]; }
        results[f].n_inliers = ints[5 * f];
        if (n) memcpy(results[f].outlier, S.host + o_flags + foff[f], n);
    }
    return SLAMIT_OK;
}

int slamit_pose_optimize(int device, const slamit_pose_problem* prob, slamit_pose_result* res) {
    return slamit_pose_optimize_batch(device, 1, prob, res);
}

}  // extern "C"
