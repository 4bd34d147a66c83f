// sim3.hip — Optimizer::OptimizeSim3 on the GPU (include/slamit.h, slamit_sim3_*).
//
// Reference: ORB_SLAM2/src/Optimizer.cc:1046-1247 driving g2o (BlockSolverX + LinearSolverDense + Levenberg) over one
// VertexSim3Expmap and, per correspondence, the pair EdgeSim3ProjectXYZ / EdgeInverseSim3ProjectXYZ
// (Thirdparty/g2o/g2o/types/types_seven_dof_expmap.h:118-152, sim3.h:69-264).  The reference leaves the analytic
// Jacobians commented out, so g2o differentiates NUMERICALLY (core/base_binary_edge.hpp:131-200: central differences,
// delta 1e-9, through oplus = Sim3(update) * estimate); the same is done here: the 14 perturbed similarities and their
// inverses are built once per iteration, every lane evaluates its pairs at all of them.  The system is one 7x7 block, so
// the whole schedule — 5 iterations, the chi2 > th2 pruning of pairs, 10 (or 5) more iterations, the inlier count —
// runs inside ONE workgroup per problem with no host round trip; a batch is one launch.  Reductions are fixed-order.
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/slamit.h"
#include "se3_device.h"
#include "slamit_internal.h"

// Device-side record of one problem (pointers into the batch slabs; global address space in device code so that the
// accesses are global_load, not flat_load -- see ba_types.h).
#if defined(__HIP_DEVICE_COMPILE__)
#define SIM3_G __attribute__((address_space(1)))
#else
#define SIM3_G
#endif
struct Sim3Prob {
    int32_t n, fix_scale;
    double intr1[4], intr2[4], S0[8], th2;   // S = q(x, y, z, w), t, s
    const SIM3_G double* p1; const SIM3_G double* p2; const SIM3_G double* o1; const SIM3_G double* o2;
    const SIM3_G double* w1; const SIM3_G double* w2;
    SIM3_G double* chi12; SIM3_G double* chi21;      // n each: chi2 of the last evaluated trial
    SIM3_G uint8_t* inlier;                          // n out
    SIM3_G double* out;                              // 16: R (9), t (3), s, chi2[2], pad
    SIM3_G int32_t* ints;                            // 4: n_inliers, n_its[2], returned-at-the-10-pair-test flag
};

namespace {

__device__ __forceinline__ double block_sum(double v, double* sh) {  // 256 threads, result in every thread
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

__device__ __forceinline__ void quat_mul(const double* a, const double* b, double* r) {   // Eigen quaternion product, (x, y, z, w)
    r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}

// Sim3(const Vector7d& update) (sim3.h:69-146) followed by (update) * S (sim3.h:258-264): S <- exp(u) * S
__device__ void sim3_oplus(double* S, const double* u_in, bool fix_scale) {
    double u[7];
    for (int i = 0; i < 7; ++i) u[i] = u_in[i];
    if (fix_scale) u[6] = 0;
    const double sigma = u[6];
    const double theta = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    const double O[9] = {0, -u[2], u[1], u[2], 0, -u[0], -u[1], u[0], 0};
    double O2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    const double es = exp(sigma);
    const double eps = 0.00001;
    double A, B, C, ca = 1.0, cb = 1.0;   // R = I + ca Omega + cb Omega^2
    if (fabs(sigma) < eps) {
        C = 1;
        if (theta < eps) { A = 1. / 2.; B = 1. / 6.; }
        else {
            const double theta2 = theta * theta;
            A = (1 - cos(theta)) / (theta2);
            B = (theta - sin(theta)) / (theta2 * theta);
            ca = sin(theta) / theta; cb = (1 - cos(theta)) / (theta * theta);
        }
    } else {
        C = (es - 1) / sigma;
        if (theta < eps) {
            const double sigma2 = sigma * sigma;
            A = ((sigma - 1) * es + 1) / sigma2;
            B = ((0.5 * sigma2 - sigma + 1) * es) / (sigma2 * sigma);
        } else {
            ca = sin(theta) / theta; cb = (1 - cos(theta)) / (theta * theta);
            const double a = es * sin(theta), b = es * cos(theta), theta2 = theta * theta, sigma2 = sigma * sigma;
            const double c = theta2 + sigma2;
            A = (a * sigma + (1 - b) * theta) / (theta * c);
            B = (C - ((b - 1) * sigma + a * theta) / (c)) * 1. / (theta2);
        }
    }
    double R[9], qe[4], te[3];
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0 ? 1.0 : 0.0) + ca * O[i] + cb * O2[i];
    R_to_quat(R, qe);   // Quaterniond(R): not normalised
    for (int i = 0; i < 3; ++i) {
        double acc = 0;
        for (int j = 0; j < 3; ++j) acc += (A * O[3 * i + j] + B * O2[3 * i + j] + (i == j ? C : 0.0)) * u[3 + j];
        te[i] = acc;
    }
    double nq[4], rt[3];
    quat_mul(qe, S, nq);
    quat_rot(qe, S + 4, rt);
    for (int i = 0; i < 4; ++i) S[i] = nq[i];
    for (int i = 0; i < 3; ++i) S[4 + i] = es * rt[i] + te[i];
    S[7] = es * S[7];
}

__device__ __forceinline__ void sim3_inverse(const double* S, double* I) {   // Sim3(r.conjugate(), r.conjugate() * ((-1. / s) * t), 1. / s)
    I[0] = -S[0]; I[1] = -S[1]; I[2] = -S[2]; I[3] = S[3];
    const double k = -1. / S[7];
    const double kt[3] = {k * S[4], k * S[5], k * S[6]};
    quat_rot(I, kt, I + 4);
    I[7] = 1. / S[7];
}

// the two edge errors of one pair at (S, S^-1) (types_seven_dof_expmap.h:128-133, 146-151)
__device__ __forceinline__ void pair_error(const double* S, const double* Si, const double* in1, const double* in2, const double* p1,
                                           const double* p2, const double* o1, const double* o2, double* e12, double* e21) {
    double r[3];
    quat_rot(S, p2, r);
    const double x = S[7] * r[0] + S[4], y = S[7] * r[1] + S[5], z = S[7] * r[2] + S[6];
    e12[0] = o1[0] - (x / z * in1[0] + in1[2]);
    e12[1] = o1[1] - (y / z * in1[1] + in1[3]);
    quat_rot(Si, p1, r);
    const double xi = Si[7] * r[0] + Si[4], yi = Si[7] * r[1] + Si[5], zi = Si[7] * r[2] + Si[6];
    e21[0] = o2[0] - (xi / zi * in2[0] + in2[2]);
    e21[1] = o2[1] - (yi / zi * in2[1] + in2[3]);
}

__device__ __forceinline__ double huber(double c2, double delta, double dsqr) { return c2 > dsqr ? 2 * sqrt(c2) * delta - dsqr : c2; }

// errors + chi2 of every active pair at S; returns the robust cost
__device__ double sim3_errors(const Sim3Prob& P, const double* S, const uint8_t* active, double delta, double* sSi, double* sh) {
    if (threadIdx.x == 0) sim3_inverse(S, sSi);
    __syncthreads();
    const double dsqr = (double)(float)(delta * delta);   // RobustKernelHuber::dsqr is a float member (g2o/core/robust_kernel_impl.h:84)
    double part = 0;
    for (int k = threadIdx.x; k < P.n; k += 256) {
        if (!active[k]) continue;
        const double p1[3] = {P.p1[3 * k], P.p1[3 * k + 1], P.p1[3 * k + 2]}, p2[3] = {P.p2[3 * k], P.p2[3 * k + 1], P.p2[3 * k + 2]};
        const double o1[2] = {P.o1[2 * k], P.o1[2 * k + 1]}, o2[2] = {P.o2[2 * k], P.o2[2 * k + 1]};
        double e12[2], e21[2];
        pair_error(S, sSi, P.intr1, P.intr2, p1, p2, o1, o2, e12, e21);
        const double w1 = P.w1[k], w2 = P.w2[k];
        const double c12 = e12[0] * w1 * e12[0] + e12[1] * w1 * e12[1], c21 = e21[0] * w2 * e21[0] + e21[1] * w2 * e21[1];
        P.chi12[k] = c12; P.chi21[k] = c21;
        part += huber(c12, delta, dsqr) + huber(c21, delta, dsqr);
    }
    return block_sum(part, sh);
}

// 7x7 LDLt without pivoting (H + lambda I) x = b; false on a zero pivot
__device__ bool solve7(const double* H, double lambda, const double* b, double* x) {
    double A[49];
    for (int i = 0; i < 49; ++i) A[i] = H[i] + (i % 8 == 0 ? lambda : 0.0);
    for (int j = 0; j < 7; ++j) {
        double d = A[8 * j];
        for (int k = 0; k < j; ++k) d -= A[7 * j + k] * A[7 * j + k] * A[8 * k];
        if (d == 0.0 || !(fabs(d) <= DBL_MAX)) return false;
        A[8 * j] = d;
        for (int i = j + 1; i < 7; ++i) {
            double s = A[7 * i + j];
            for (int k = 0; k < j; ++k) s -= A[7 * i + k] * A[7 * j + k] * A[8 * k];
            A[7 * i + j] = s / d;
        }
    }
    for (int i = 0; i < 7; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= A[7 * i + k] * x[k]; x[i] = s; }
    for (int i = 0; i < 7; ++i) x[i] /= A[8 * i];
    for (int i = 6; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k < 7; ++k) s -= A[7 * k + i] * x[k]; x[i] = s; }
    return true;
}

}  // namespace

__global__ __launch_bounds__(256) void sim3_opt_kernel(const Sim3Prob* probs) {
    const Sim3Prob P = probs[blockIdx.x];
    const int tid = threadIdx.x, n = P.n;
    __shared__ double sh[4];
    __shared__ double sS[8], sSbak[8], sSi[8], sPert[28][8];   // sPert: [2d] = S(+delta e_d), [2d+1] = S(-delta e_d), [14 + ..] their inverses
    __shared__ double sH[49], sb[7], sx[7];
    __shared__ double s_lambda, s_ni, s_cur, s_rho;
    __shared__ int s_ok2, s_cnt;
    extern __shared__ uint8_t s_act[];   // n bytes: the pair is still in the graph
    const float th2f = (float)P.th2;
    const double th2 = (double)th2f;
    const double delta = (double)sqrtf(th2f);   // const float deltaHuber = sqrt(th2)
    const bool fix = P.fix_scale != 0;
    if (tid < 8) sS[tid] = P.S0[tid];
    for (int k = tid; k < n; k += 256) { s_act[k] = 1; P.inlier[k] = 1; P.chi12[k] = 0; P.chi21[k] = 0; }
    if (tid == 0) { P.ints[0] = 0; P.ints[1] = 0; P.ints[2] = 0; P.ints[3] = 0; P.out[13] = 0; P.out[14] = 0; }
    __syncthreads();
    int nBadPairs = 0;
    bool early = false;
    for (int stage = 0; stage < 2 && !early; ++stage) {
        const int iterations = stage == 0 ? 5 : (nBadPairs > 0 ? 10 : 5);
        int nact = 0;
        for (int k = tid; k < n; k += 256) nact += s_act[k];
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        if (nact) atomicAdd(&s_cnt, nact);
        __syncthreads();
        const bool any_active = s_cnt > 0;
        __syncthreads();
        int done = 0;
        double lastChi = 0;
        if (any_active) {
            int lm_nBad = 0;
            bool ok = true;
            for (int it = 0; it < iterations && ok; ++it) {
                // g2o re-evaluates the errors at the top of every iteration; after an accepted trial (the only way to get here
                // with it > 0) errors, chi2 and S^-1 are the ones that trial just computed at this very S: one pass saved
                const double currentChi0 = it == 0 ? sim3_errors(P, sS, s_act, delta, sSi, sh) : s_cur;
                // the 14 perturbed similarities of g2o's numeric differentiation and their inverses
                if (tid < 14) {
                    double Sp[8], add[7] = {0, 0, 0, 0, 0, 0, 0};
                    for (int i = 0; i < 8; ++i) Sp[i] = sS[i];
                    add[tid >> 1] = (tid & 1) ? -1e-9 : 1e-9;
                    sim3_oplus(Sp, add, fix);
                    double Ip[8];
                    sim3_inverse(Sp, Ip);
                    for (int i = 0; i < 8; ++i) { sPert[tid][i] = Sp[i]; sPert[14 + tid][i] = Ip[i]; }
                }
                __syncthreads();
                // ---- normal equations H (28 unique), b (7): per-thread partials, shuffle tree, 4 waves in order ----
                double h[28], bb[7];
                for (int i = 0; i < 28; ++i) h[i] = 0;
                for (int i = 0; i < 7; ++i) bb[i] = 0;
                const double dsqr = (double)(float)(delta * delta), scalar = 1.0 / (2 * 1e-9);
                for (int k = tid; k < n; k += 256) {
                    if (!s_act[k]) continue;
                    const double p1[3] = {P.p1[3 * k], P.p1[3 * k + 1], P.p1[3 * k + 2]}, p2[3] = {P.p2[3 * k], P.p2[3 * k + 1], P.p2[3 * k + 2]};
                    const double o1[2] = {P.o1[2 * k], P.o1[2 * k + 1]}, o2[2] = {P.o2[2 * k], P.o2[2 * k + 1]};
                    double J12[14], J21[14];
#pragma unroll
                    for (int d = 0; d < 7; ++d) {
                        double a12[2], a21[2], c12[2], c21[2];
                        pair_error(sPert[2 * d], sPert[14 + 2 * d], P.intr1, P.intr2, p1, p2, o1, o2, a12, a21);
                        pair_error(sPert[2 * d + 1], sPert[14 + 2 * d + 1], P.intr1, P.intr2, p1, p2, o1, o2, c12, c21);
                        J12[d] = scalar * (a12[0] - c12[0]); J12[7 + d] = scalar * (a12[1] - c12[1]);
                        J21[d] = scalar * (a21[0] - c21[0]); J21[7 + d] = scalar * (a21[1] - c21[1]);
                    }
                    double e12[2], e21[2];
                    pair_error(sS, sSi, P.intr1, P.intr2, p1, p2, o1, o2, e12, e21);
                    const double w1 = P.w1[k], w2 = P.w2[k], c12 = P.chi12[k], c21 = P.chi21[k];
                    const double r12 = c12 > dsqr ? delta / sqrt(c12) : 1.0, r21 = c21 > dsqr ? delta / sqrt(c21) : 1.0;
                    const double wa = r12 * w1, wb = r21 * w2;
                    int q = 0;
#pragma unroll
                    for (int a = 0; a < 7; ++a) {
                        bb[a] -= r12 * (J12[a] * w1 * e12[0] + J12[7 + a] * w1 * e12[1]) + r21 * (J21[a] * w2 * e21[0] + J21[7 + a] * w2 * e21[1]);
#pragma unroll
                        for (int c = a; c < 7; ++c)
                            h[q++] += (J12[a] * J12[c] + J12[7 + a] * J12[7 + c]) * wa + (J21[a] * J21[c] + J21[7 + a] * J21[7 + c]) * wb;
                    }
                }
                __shared__ double red[4][35];
                for (int i = 0; i < 28; ++i) { const double v = wave_sum(h[i]); if ((tid & 63) == 0) red[tid >> 6][i] = v; }
                for (int i = 0; i < 7; ++i) { const double v = wave_sum(bb[i]); if ((tid & 63) == 0) red[tid >> 6][28 + i] = v; }
                __syncthreads();
                if (tid == 0) {
                    int q = 0;
                    for (int a = 0; a < 7; ++a)
                        for (int c = a; c < 7; ++c) { const double v = red[0][q] + red[1][q] + red[2][q] + red[3][q]; sH[7 * a + c] = v; sH[7 * c + a] = v; ++q; }
                    for (int a = 0; a < 7; ++a) sb[a] = red[0][28 + a] + red[1][28 + a] + red[2][28 + a] + red[3][28 + a];
                    if (it == 0) {
                        double m = 0;
                        for (int j = 0; j < 7; ++j) m = fmax(m, fabs(sH[8 * j]));
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
                        for (int i = 0; i < 8; ++i) sSbak[i] = sS[i];
                        double x[7];
                        const bool ok2 = solve7(sH, s_lambda, sb, x);
                        if (ok2) { double T[8]; for (int i = 0; i < 8; ++i) T[i] = sS[i]; sim3_oplus(T, x, fix); for (int i = 0; i < 8; ++i) sS[i] = T[i]; }
                        else for (int i = 0; i < 7; ++i) x[i] = 0;
                        for (int i = 0; i < 7; ++i) sx[i] = x[i];
                        s_ok2 = ok2;
                    }
                    __syncthreads();
                    tempChi = sim3_errors(P, sS, s_act, delta, sSi, sh);
                    if (!s_ok2) tempChi = DBL_MAX;
                    if (tid == 0) {
                        double scale = 0;
                        for (int k = 0; k < 7; ++k) scale += sx[k] * (s_lambda * sx[k] + sb[k]);
                        const double r = (s_cur - tempChi) / (scale + 1e-3);
                        if (r > 0 && fabs(tempChi) <= DBL_MAX) {
                            const double alpha = fmin(1. - pow((2 * r - 1), 3), 2. / 3.);
                            s_lambda *= fmax(1. / 3., alpha);
                            s_ni = 2; s_cur = tempChi;
                        } else {
                            s_lambda *= s_ni; s_ni *= 2;
                            for (int i = 0; i < 8; ++i) sS[i] = sSbak[i];
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
        if (tid == 0) { P.ints[1 + stage] = done; P.out[13 + stage] = lastChi; }
        // ---- the chi2 tests (:1184-1201, 1218-1234): the chi2 of the LAST EVALUATED trial, accepted or not ----
        int bad = 0;
        for (int k = tid; k < n; k += 256) {
            if (!s_act[k]) continue;
            if (P.chi12[k] > th2 || P.chi21[k] > th2) {
                P.inlier[k] = 0;
                if (stage == 0) s_act[k] = 0;
                ++bad;
            }
        }
        if (tid == 0) s_cnt = 0;
        __syncthreads();
        if (bad) atomicAdd(&s_cnt, bad);
        __syncthreads();
        const int nbad = s_cnt;
        __syncthreads();
        if (stage == 0) {
            nBadPairs = nbad;
            if (n - nBadPairs < 10) {   // :1212-1213: return 0, g2oS12 untouched
                early = true;
                if (tid == 0) {
                    double R[9];
                    quat_to_R(P.S0, R);
                    for (int i = 0; i < 9; ++i) P.out[i] = R[i];
                    for (int i = 0; i < 3; ++i) P.out[9 + i] = P.S0[4 + i];
                    P.out[12] = P.S0[7];
                    P.ints[0] = 0; P.ints[3] = 1;
                }
            }
        } else if (tid == 0) {
            double R[9];
            quat_to_R(sS, R);
            for (int i = 0; i < 9; ++i) P.out[i] = R[i];
            for (int i = 0; i < 3; ++i) P.out[9 + i] = sS[4 + i];
            P.out[12] = sS[7];
            P.ints[0] = n - nBadPairs - nbad;
        }
    }
}

extern "C" {

int slamit_sim3_optimize_batch(int device, int nprob, const slamit_sim3_problem* probs, slamit_sim3_result* results) {
    if (nprob < 0 || (nprob && (!probs || !results))) return slamit_fail(SLAMIT_ERR_ARG, "slamit_sim3_optimize_batch: bad argument");
    if (nprob == 0) return SLAMIT_OK;
    size_t total = 0, flag_total = 0;
    int nmax = 1;
    std::vector<size_t> off(nprob), foff(nprob);
    for (int f = 0; f < nprob; ++f) {
        const slamit_sim3_problem& P = probs[f];
        if (P.n < 0 || (P.n && (!P.p1 || !P.p2 || !P.obs1 || !P.obs2 || !P.inv_sigma2_1 || !P.inv_sigma2_2 || !results[f].inlier)))
            return slamit_fail(SLAMIT_ERR_ARG, "slamit_sim3_optimize_batch: null array");
        if (!(P.s12 > 0) || !(P.th2 > 0)) return slamit_fail(SLAMIT_ERR_ARG, "slamit_sim3_optimize_batch: scale and th2 must be positive");
        off[f] = total; foff[f] = flag_total;
        total += (size_t)14 * P.n + 16;          // p1 3n | p2 3n | o1 2n | o2 2n | w1 n | w2 n | chi12 n | chi21 n | out 16   (doubles)
        flag_total += ((size_t)P.n + 15) & ~(size_t)7;
        nmax = std::max(nmax, (int)P.n);
    }
    SLAMIT_USE_DEVICE(device);
    // one slab per host thread (slamit_internal.h): [doubles of every problem | ints | Sim3Prob records | flags], one copy each way
    const size_t o_ints = sizeof(double) * total, o_probs = (o_ints + sizeof(int32_t) * 4 * nprob + 15) & ~(size_t)15;
    const size_t o_flags = o_probs + sizeof(Sim3Prob) * nprob, bytes = o_flags + flag_total;
    static thread_local SlamitScratch S;
    hipError_t e = slamit_scratch_reserve(S, device, bytes);
    if (e != hipSuccess) return slamit_fail_hip(e, "slamit_sim3_optimize_batch");
    double* stage = reinterpret_cast<double*>(S.host);
    double* d_buf = reinterpret_cast<double*>(S.dev);
    int32_t* d_ints = reinterpret_cast<int32_t*>(S.dev + o_ints);
    Sim3Prob* pr = reinterpret_cast<Sim3Prob*>(S.host + o_probs);
    uint8_t* d_flags = S.dev + o_flags;
    for (int f = 0; f < nprob; ++f) {
        const slamit_sim3_problem& P = probs[f];
        double* h = stage + off[f];
        double* d = d_buf + off[f];
        const size_t n = P.n;
        if (n) {
            memcpy(h, P.p1, 24 * n); memcpy(h + 3 * n, P.p2, 24 * n); memcpy(h + 6 * n, P.obs1, 16 * n); memcpy(h + 8 * n, P.obs2, 16 * n);
            memcpy(h + 10 * n, P.inv_sigma2_1, 8 * n); memcpy(h + 11 * n, P.inv_sigma2_2, 8 * n);
        }
        Sim3Prob& Q = pr[f];
        memset(&Q, 0, sizeof(Q));
        Q.n = P.n; Q.fix_scale = P.fix_scale; Q.th2 = P.th2;
        memcpy(Q.intr1, P.intr1, sizeof(Q.intr1)); memcpy(Q.intr2, P.intr2, sizeof(Q.intr2));
        {   // Sim3(R, t, s): Quaterniond(R) by Eigen's rule, not normalised (host copy of se3_device.h:R_to_quat)
            const double* m = P.r12;
            double* q = Q.S0;
            double t = m[0] + m[4] + m[8];
            if (t > 0) {
                t = sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
                q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
            } else {
                int i = 0;
                if (m[4] > m[0]) i = 1;
                if (m[8] > m[4 * i]) i = 2;
                const int j = (i + 1) % 3, k = (j + 1) % 3;
                t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
                q[i] = 0.5 * t; t = 0.5 / t;
                q[3] = (m[3 * k + j] - m[3 * j + k]) * t; q[j] = (m[3 * j + i] + m[3 * i + j]) * t; q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
            }
            for (int i = 0; i < 3; ++i) Q.S0[4 + i] = P.t12[i];
            Q.S0[7] = P.s12;
        }
        typedef SIM3_G double gd;
        Q.p1 = (const gd*)d; Q.p2 = (const gd*)(d + 3 * n); Q.o1 = (const gd*)(d + 6 * n); Q.o2 = (const gd*)(d + 8 * n);
        Q.w1 = (const gd*)(d + 10 * n); Q.w2 = (const gd*)(d + 11 * n); Q.chi12 = (gd*)(d + 12 * n); Q.chi21 = (gd*)(d + 13 * n);
        Q.out = (gd*)(d + 14 * n);
        Q.inlier = (SIM3_G uint8_t*)(d_flags + foff[f]);
        Q.ints = (SIM3_G int32_t*)(d_ints + 4 * f);
    }
    e = hipMemcpyAsync(S.dev, S.host, o_flags, hipMemcpyHostToDevice, S.st);
    if (e == hipSuccess) {
        if (nmax > 32 * 1024) e = hipFuncSetAttribute(reinterpret_cast<const void*>(sim3_opt_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, nmax + 16);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(sim3_opt_kernel, dim3(nprob), dim3(256), (size_t)nmax + 16, S.st, reinterpret_cast<const Sim3Prob*>(S.dev + o_probs));
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(S.host, S.dev, bytes, hipMemcpyDeviceToHost, S.st);
    if (e == hipSuccess) e = hipStreamSynchronize(S.st);
    if (e != hipSuccess) return slamit_fail_hip(e, "slamit_sim3_optimize_batch");
    const int32_t* ints = reinterpret_cast<const int32_t*>(S.host + o_ints);
    for (int f = 0; f < nprob; ++f) {
        const size_t n = probs[f].n;
        const double* o = stage + off[f] + 14 * n;
        memcpy(results[f].r12, o, 72); memcpy(results[f].t12, o + 9, 24);
        results[f].s12 = o[12];
        results[f].chi2[0] = o[13]; results[f].chi2[1] = o[14];
        results[f].n_inliers = ints[4 * f]; results[f].n_its[0] = ints[4 * f + 1]; results[f].n_its[1] = ints[4 * f + 2];
        if (ints[4 * f + 3]) {   // the reference returned before touching g2oS12: hand the input back bit for bit
            memcpy(results[f].r12, probs[f].r12, 72); memcpy(results[f].t12, probs[f].t12, 24);
            results[f].s12 = probs[f].s12;
        }
        if (n) memcpy(results[f].inlier, S.host + o_flags + foff[f], n);
    }
    return SLAMIT_OK;
}

int slamit_sim3_optimize(int device, const slamit_sim3_problem* prob, slamit_sim3_result* res) {
    return slamit_sim3_optimize_batch(device, 1, prob, res);
}

}  // extern "C"
