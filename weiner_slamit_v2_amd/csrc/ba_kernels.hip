// ba_kernels.hip — hand-written gfx950 kernels of the local bundle adjustment (fp64).
//
// What they replace (reference: ORB_SLAM2/src/Optimizer.cc:453-778 driving g2o; "G/" =
// Thirdparty/g2o/g2o/):
//   k_linearize        EdgeSE3ProjectXYZ::computeError/linearizeOplus  G/types/types_six_dof_expmap.{h:90-95,cpp:103-139}
//                      + Huber weight                                   G/core/robust_kernel_impl.cpp:77-90
//   k_point_reduce     constructQuadraticForm, landmark side            G/core/base_binary_edge.hpp:55-120
//   k_pose_reduce      constructQuadraticForm, pose side
//   k_iter_begin       computeLambdaInit                                G/core/optimization_algorithm_levenberg.cpp:93-97,166-180
//   k_prepare          setLambda + Dinv + the Schur operand Hpl Ci^T      G/core/block_solver.hpp:564-589,381-400
//   k_schur / _reduce  Schur complement  S = Hpp - sum_p B Dinv B^T      G/core/block_solver.hpp:401-439
//                      as the product G G^T of ONE operand (G = Hpl Ci^T, point_chol) on v_mfma_f64_16x16x4_f64
//   k_point_pass       k_linearize + k_point_reduce + k_prepare in one launch: every LM trial after a stage's first
//   k_schur_pose       k_schur with k_pose_reduce's workgroups behind its tiles: the same trials
//   k_ldlt_band / k_ldlt_blocked   LinearSolverEigen::solve              G/solvers/linear_solver_eigen.h:94-124
//                      banded systems: block LDLt with 4 x 4 pivots inside LDS, rank-4 updates on v_mfma_f64_16x16x4_f64 with
//                      the tiles resident in accumulators (ldlt_band_solve); others: dense blocked LDLt through L2
//   k_backsub_update   landmark back-substitution + oplus + push()       G/core/block_solver.hpp:459-485, sparse_optimizer.cpp:422-435
//   k_errors           computeActiveErrors + activeRobustChi2            G/core/sparse_optimizer.cpp:61-114
//   k_decide           gain ratio, lambda update, pop()/discardTop(), stop rules   ...levenberg.cpp:102-161
//   k_gate / k_final   chi2 gate + depth test between / after the stages  Optimizer.cc:672-743
//
// The whole Levenberg-Marquardt control flow lives in a device-resident BaState per window, so
// an LM trial is a fixed sequence of launches ("slot": eleven for the first trial of a stage, seven afterwards,
// bak_slot) with no host round trip; kernels of a finished window exit at once.  blockIdx.y is the window of a batch.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <float.h>
#include <stdint.h>

#define BA_GLOBAL_POINTERS   // BaWin members are global-address-space pointers in this file's device code
#include "ba_types.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
// A pointer read out of BaWin (a struct in HBM) is a GENERIC pointer to the compiler: every access through it is a
// flat_load / flat_store, which also counts in lgkmcnt -- so each LDS wait of a software pipeline waits for the
// prefetched HBM data as well.  The hot kernels cast their matrices to the global address space once.
typedef __attribute__((address_space(1))) double gdouble;
typedef double double2_t __attribute__((ext_vector_type(2)));

#include "se3_device.h"

// deterministic block sum for 256-thread blocks; result valid in thread 0
__device__ double block_sum_256(double v, double* sh /*[4]*/) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

// The LM state of window b sits at a FIXED place relative to the window table -- states are laid out in reverse order right in front of
// it (ba_api.hip) --, so a kernel's first load of `done` does not wait for the BaWin load that would hand it the pointer: one dependent
// global round trip (~1 us) less at the head of every launch of an LM slot.
#define BA_ST(wins, b) (reinterpret_cast<BaState*>(wins) - ((int)(b) + 1))

// ---- per-edge geometry -----------------------------------------------------------------------
struct EdgeGeom { double err0, err1, chi2, x, y, z; };

__device__ __forceinline__ EdgeGeom edge_eval(const BaWin& W, int e) {
    const int kf = W.e_kf[e], pt = W.e_pt[e];
    const double* T = W.pose + 7 * kf;
    double Xc[3];
    quat_rot(T, W.pt + 3 * pt, Xc);
    Xc[0] += T[4]; Xc[1] += T[5]; Xc[2] += T[6];
    const double* in = W.intr + 4 * kf;
    EdgeGeom g;
    g.x = Xc[0]; g.y = Xc[1]; g.z = Xc[2];
    g.err0 = W.e_uv[2 * e] - (Xc[0] / Xc[2] * in[0] + in[2]);
    g.err1 = W.e_uv[2 * e + 1] - (Xc[1] / Xc[2] * in[1] + in[3]);
    const double w = W.e_w[e];
    g.chi2 = g.err0 * w * g.err0 + g.err1 * w * g.err1;
    return g;
}

// RobustKernelHuber keeps delta^2 in a FLOAT member (g2o/core/robust_kernel_impl.h:84, written by setDelta, robust_kernel_impl.cpp:64-68):
// the inlier test and the constant of rho(e) = 2 delta sqrt(e) - delta^2 use the rounded value
__device__ __forceinline__ double huber_dsqr(double delta) { return (double)(float)(delta * delta); }

// ---- windows with stereo observations (EdgeStereoSE3ProjectXYZ, g2o/types/types_six_dof_expmap.h:112-141): three residual rows
// per edge record (BA_JAC_STEREO doubles: A 3x3 | B 3x6 | wO | r0 r1 r2); a monocular edge of such a window keeps its own
// expressions and a zero third row.  The monocular code above and below is untouched: a window takes one path or the other.
struct EdgeGeom3 { double err0, err1, err2, chi2, x, y, z; bool stereo; };

__device__ __forceinline__ bool edge_is_stereo(const BaWin& W, int e) { return W.nrow == 3 && !(W.e_ur[e] < 0.0); }   // Optimizer.cc:596

__device__ __forceinline__ EdgeGeom3 edge_eval3(const BaWin& W, int e) {
    const int kf = W.e_kf[e], pt = W.e_pt[e];
    const double* T = W.pose + 7 * kf;
    double Xc[3];
    quat_rot(T, W.pt + 3 * pt, Xc);
    Xc[0] += T[4]; Xc[1] += T[5]; Xc[2] += T[6];
    const double* in = W.intr + 4 * kf;
    EdgeGeom3 g;
    g.x = Xc[0]; g.y = Xc[1]; g.z = Xc[2];
    const double ur = W.e_ur[e];
    g.stereo = !(ur < 0.0);
    const double w = W.e_w[e];
    if (!g.stereo) {
        g.err0 = W.e_uv[2 * e] - (Xc[0] / Xc[2] * in[0] + in[2]);
        g.err1 = W.e_uv[2 * e + 1] - (Xc[1] / Xc[2] * in[1] + in[3]);
        g.err2 = 0.0;
        g.chi2 = g.err0 * w * g.err0 + g.err1 * w * g.err1;
    } else {
        // cam_project (types_six_dof_expmap.cpp:150-157): `const float invz = 1.0f / z` (the quotient rounded to float), bf arrives
        // through a `const float&`, and bf * invz is a FLOAT product
        const float invz = (float)(1.0 / Xc[2]);
        const double r0 = Xc[0] * (double)invz * in[0] + in[2];
        const double r1 = Xc[1] * (double)invz * in[1] + in[3];
        const double r2 = r0 - (double)__fmul_rn((float)W.bf[kf], invz);
        g.err0 = W.e_uv[2 * e] - r0; g.err1 = W.e_uv[2 * e + 1] - r1; g.err2 = ur - r2;
        g.chi2 = g.err0 * w * g.err0 + g.err1 * w * g.err1 + g.err2 * w * g.err2;
    }
    return g;
}

__device__ __forceinline__ void edge_jacobian3(const BaWin& W, bool robust, int e, double* J /*[31]*/) {
    const EdgeGeom3 g = edge_eval3(W, e);
    const int kf = W.e_kf[e];
    const double* in = W.intr + 4 * kf;
    const double fx = in[0], fy = in[1];
    double R[9];
    quat_to_R(W.pose + 7 * kf, R);
    const double x = g.x, y = g.y, z = g.z, z_2 = z * z;
    double delta = W.huber_delta;
    if (!g.stereo) {   // EdgeSE3ProjectXYZ::linearizeOplus, as in edge_jacobian
        const double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
        for (int r = 0; r < 2; ++r)
            for (int c = 0; c < 3; ++c)
                J[3 * r + c] = -1. / z * (tmp[3 * r] * R[c] + tmp[3 * r + 1] * R[3 + c] + tmp[3 * r + 2] * R[6 + c]);
        J[6] = 0; J[7] = 0; J[8] = 0;
    } else {           // EdgeStereoSE3ProjectXYZ::linearizeOplus (types_six_dof_expmap.cpp:188-234), its expressions as written
        const double bf = W.bf[kf];
        for (int c = 0; c < 3; ++c) {
            J[c] = -fx * R[c] / z + fx * x * R[6 + c] / z_2;
            J[3 + c] = -fy * R[3 + c] / z + fy * y * R[6 + c] / z_2;
            J[6 + c] = J[c] - bf * R[6 + c] / z_2;
        }
        delta = W.huber_delta_s;
    }
    double* B = J + 9;
    B[0] = x * y / z_2 * fx; B[1] = -(1 + (x * x / z_2)) * fx; B[2] = y / z * fx;
    B[3] = -1. / z * fx; B[4] = 0; B[5] = x / z_2 * fx;
    B[6] = (1 + y * y / z_2) * fy; B[7] = -x * y / z_2 * fy; B[8] = -x / z * fy;
    B[9] = 0; B[10] = -1. / z * fy; B[11] = y / z_2 * fy;
    if (g.stereo) {
        const double bf = W.bf[kf];
        B[12] = B[0] - bf * y / z_2; B[13] = B[1] + bf * x / z_2; B[14] = B[2];
        B[15] = B[3]; B[16] = 0; B[17] = B[5] - bf / z_2;
    } else {
        for (int i = 12; i < 18; ++i) B[i] = 0;
    }
    double rho1 = 1.0;
    if (robust && g.chi2 > huber_dsqr(delta)) rho1 = delta / sqrt(g.chi2);
    const double w = W.e_w[e];
    J[27] = rho1 * w;
    J[28] = -w * g.err0 * rho1;
    J[29] = -w * g.err1 * rho1;
    J[30] = -w * g.err2 * rho1;
}

// one edge's share of its point's blocks: b[3] += A^T omega_r, h[6] (xx xy xz yy yz zz) += A^T wO A
__device__ __forceinline__ void point_accum3(const double* J, double* h, double* b) {
    const double wO = J[27], r0 = J[28], r1 = J[29], r2 = J[30];
    b[0] += J[0] * r0 + J[3] * r1 + J[6] * r2; b[1] += J[1] * r0 + J[4] * r1 + J[7] * r2; b[2] += J[2] * r0 + J[5] * r1 + J[8] * r2;
    h[0] += (J[0] * J[0] + J[3] * J[3] + J[6] * J[6]) * wO; h[1] += (J[0] * J[1] + J[3] * J[4] + J[6] * J[7]) * wO; h[2] += (J[0] * J[2] + J[3] * J[5] + J[6] * J[8]) * wO;
    h[3] += (J[1] * J[1] + J[4] * J[4] + J[7] * J[7]) * wO; h[4] += (J[1] * J[2] + J[4] * J[5] + J[7] * J[8]) * wO; h[5] += (J[2] * J[2] + J[5] * J[5] + J[8] * J[8]) * wO;
}

// Hpl (6 x 3) of one edge of a stereo window
__device__ __forceinline__ void hpl_of3(const double* J, double* H /*6x3*/) {
    const double wO = J[27];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) H[3 * i + j] = (J[9 + i] * J[j] + J[15 + i] * J[3 + j] + J[21 + i] * J[6 + j]) * wO;
}

// ---- S1: Jacobians + weights of every active edge --------------------------------------------
__device__ __forceinline__ void edge_jacobian(const BaWin& W, bool robust, int e, double* J /*[21]*/) {
    const EdgeGeom g = edge_eval(W, e);
    const int kf = W.e_kf[e];
    const double* in = W.intr + 4 * kf;
    const double fx = in[0], fy = in[1];
    double R[9];
    quat_to_R(W.pose + 7 * kf, R);
    const double x = g.x, y = g.y, z = g.z, z_2 = z * z;
    const double tmp[6] = {fx, 0, -x / z * fx, 0, fy, -y / z * fy};
    for (int r = 0; r < 2; ++r)
        for (int c = 0; c < 3; ++c)
            J[3 * r + c] = -1. / z * (tmp[3 * r] * R[c] + tmp[3 * r + 1] * R[3 + c] + tmp[3 * r + 2] * R[6 + c]);
    J[6] = x * y / z_2 * fx; J[7] = -(1 + (x * x / z_2)) * fx; J[8] = y / z * fx;
    J[9] = -1. / z * fx; J[10] = 0; J[11] = x / z_2 * fx;
    J[12] = (1 + y * y / z_2) * fy; J[13] = -x * y / z_2 * fy; J[14] = -x / z * fy;
    J[15] = 0; J[16] = -1. / z * fy; J[17] = y / z_2 * fy;
    const double dsqr = huber_dsqr(W.huber_delta);
    double rho1 = 1.0;
    if (robust && g.chi2 > dsqr) rho1 = W.huber_delta / sqrt(g.chi2);
    const double w = W.e_w[e];
    J[18] = rho1 * w;                 // weightedOmega
    J[19] = -w * g.err0 * rho1;       // omega_r
    J[20] = -w * g.err1 * rho1;
}

__global__ __launch_bounds__(256) void k_linearize(BaWin* wins) {
    const BaWin& W = wins[blockIdx.y];
    BaState* st = BA_ST(wins, blockIdx.y);
    if (st->done || !st->need_linearize) return;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e == 0 && st->it == 0) st->maxdiag_bits = 0ull;
    if (e >= W.n_edge || !W.e_active[e]) return;
    if (W.nrow == 3) {
        double J3[BA_JAC_STEREO];
        edge_jacobian3(W, st->robust != 0, e, J3);
        double* Jm = W.e_jac + BA_JAC_STEREO * (size_t)e;
#pragma unroll
        for (int i = 0; i < BA_JAC_STEREO; ++i) Jm[i] = J3[i];
        return;
    }
    double Jr[21];
    edge_jacobian(W, st->robust != 0, e, Jr);
    double* J = W.e_jac + 21 * (size_t)e;
#pragma unroll
    for (int i = 0; i < 21; ++i) J[i] = Jr[i];
}

__device__ __forceinline__ void atomic_max_bits(unsigned long long* p, double v) {
    atomicMax(p, (unsigned long long)__double_as_longlong(fabs(v)));
}

// ---- S2: landmark blocks Hll, bl: BA_PG lanes per point (lane g takes edges g, g + BA_PG, ... of the point's list, then a
// fixed xor-shuffle tree), so a 50-observation point is 7 serial edges instead of 50 and a wave covers 8 points -------
#define BA_PG 8
__device__ __forceinline__ double group_sum(double v) {   // sum over the BA_PG = 8 consecutive lanes of a point group
    // the xor-1 / xor-2 / xor-4 tree as three DPP exchanges (bit-identical to the __shfl_xor form, without its LDS round trips;
    // after the first two steps a quad is uniform, so the half-row mirror is as good a partner as lane ^ 4)
    v += dpp_xchg_d<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_xchg_d<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_xchg_d<0x141>(v);   // row_half_mirror
    return v;
}

__global__ __launch_bounds__(256) void k_point_reduce(BaWin* wins) {
    const BaWin& W = wins[blockIdx.y];
    BaState* st = BA_ST(wins, blockIdx.y);
    if (st->done || !st->need_linearize) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int p = t / BA_PG, g = t % BA_PG;
    const bool live = p < W.n_pt;
    double h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    if (live)
        for (int i = W.pt_ptr[p] + g; i < W.pt_ptr[p + 1]; i += BA_PG) {
            const int e = W.pt_edges[i];
            if (!W.e_active[e]) continue;
            if (W.nrow == 3) { point_accum3(W.e_jac + BA_JAC_STEREO * (size_t)e, h, b); continue; }
            const double* J = W.e_jac + 21 * (size_t)e;
            const double wO = J[18], r0 = J[19], r1 = J[20];
            b[0] += J[0] * r0 + J[3] * r1; b[1] += J[1] * r0 + J[4] * r1; b[2] += J[2] * r0 + J[5] * r1;
            h[0] += (J[0] * J[0] + J[3] * J[3]) * wO; h[1] += (J[0] * J[1] + J[3] * J[4]) * wO; h[2] += (J[0] * J[2] + J[3] * J[5]) * wO;
            h[3] += (J[1] * J[1] + J[4] * J[4]) * wO; h[4] += (J[1] * J[2] + J[4] * J[5]) * wO; h[5] += (J[2] * J[2] + J[5] * J[5]) * wO;
        }
#pragma unroll
    for (int i = 0; i < 6; ++i) h[i] = group_sum(h[i]);
#pragma unroll
    for (int i = 0; i < 3; ++i) b[i] = group_sum(b[i]);
    if (!live || g != 0) return;
    for (int i = 0; i < 6; ++i) W.Hll[6 * (size_t)p + i] = h[i];
    for (int i = 0; i < 3; ++i) W.bl[3 * (size_t)p + i] = b[i];
    if (st->it == 0) {
        double m = fmax(fabs(h[0]), fmax(fabs(h[3]), fabs(h[5])));
        atomic_max_bits(&st->maxdiag_bits, m);
    }
}

// ---- S3: pose blocks Hpp, bp: one workgroup of 256 threads per keyframe.  Thread-strided edge loop, then every 16-lane row
// is summed with four DPP exchanges (no LDS round trip) and the 16 row sums of the workgroup are added in a fixed order ----
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {   // all 16 lanes of a DPP row end up with the row's sum
    v += dpp_d<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_d<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_d<0x141>(v);   // row_half_mirror
    v += dpp_d<0x140>(v);   // row_mirror
    return v;
}

__device__ __forceinline__ void pose_reduce_body(const BaWin& W, BaState* st, int kf) {
    if (st->done || !st->need_linearize) return;
    if (kf >= W.n_kf) return;
    const int col = W.pose_col[kf];
    if (col < 0) return;
    __shared__ double s_rows[27][16];
    const int tid = threadIdx.x;
    double acc[27];   // b[6] then the 21 upper-triangular entries of H
    for (int i = 0; i < 27; ++i) acc[i] = 0;
    for (int i = W.kf_ptr[kf] + tid; i < W.kf_ptr[kf + 1]; i += 256) {
        const int e = W.kf_edges[i];
        if (!W.e_active[e]) continue;
        if (W.nrow == 3) {
            const double* J = W.e_jac + BA_JAC_STEREO * (size_t)e;
            const double wO = J[27], r0 = J[28], r1 = J[29], r2 = J[30];
            int k = 6;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                acc[a] += J[9 + a] * r0 + J[15 + a] * r1 + J[21 + a] * r2;
#pragma unroll
                for (int c = a; c < 6; ++c) acc[k++] += (J[9 + a] * J[9 + c] + J[15 + a] * J[15 + c] + J[21 + a] * J[21 + c]) * wO;
            }
            continue;
        }
        const double* J = W.e_jac + 21 * (size_t)e;
        const double wO = J[18], r0 = J[19], r1 = J[20];
        int k = 6;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            acc[a] += J[6 + a] * r0 + J[12 + a] * r1;
#pragma unroll
            for (int c = a; c < 6; ++c) acc[k++] += (J[6 + a] * J[6 + c] + J[12 + a] * J[12 + c]) * wO;
        }
    }
#pragma unroll
    for (int i = 0; i < 27; ++i) {
        const double r = row16_sum(acc[i]);
        if ((tid & 15) == 0) s_rows[i][tid >> 4] = r;
    }
    __syncthreads();
    if (tid < 27) {
        double v = 0;
        for (int j = 0; j < 16; ++j) v += s_rows[tid][j];
        s_rows[tid][0] = v;
    }
    __syncthreads();
    if (tid == 0) {
        double* H = W.Hpp + 36 * (size_t)col;
        int k = 6;
        double m = 0;
        for (int a = 0; a < 6; ++a)
            for (int c = a; c < 6; ++c) { const double v = s_rows[k][0]; H[6 * a + c] = v; H[6 * c + a] = v; if (a == c) m = fmax(m, fabs(v)); ++k; }
        for (int a = 0; a < 6; ++a) W.bp[6 * (size_t)col + a] = s_rows[a][0];
        if (st->it == 0) atomic_max_bits(&st->maxdiag_bits, m);
    }
}

__global__ __launch_bounds__(256) void k_pose_reduce(BaWin* wins) { pose_reduce_body(wins[blockIdx.y], BA_ST(wins, blockIdx.y), blockIdx.x); }

// ---- S4: iteration bookkeeping (one thread per window) -----------------------------------------
__global__ void k_iter_begin(BaWin* wins) {
    const BaWin& W = wins[blockIdx.y];
    BaState* st = BA_ST(wins, blockIdx.y);
    if (threadIdx.x != 0 || st->done || !st->need_linearize) return;
    if (st->it == 0) {  // computeLambdaInit: tau * max |H_jj|
        st->lambda = 1e-5 * __longlong_as_double((long long)st->maxdiag_bits);
        st->ni = 2;
        st->nBad = 0;
    }
    st->iniChi = st->currentChi;
    st->qmax = 0;
    st->need_linearize = 0;
}

// ---- S5: damped landmark blocks, Dinv, and the two dense operands of the Schur product ----------
__device__ __forceinline__ void hpl_of(const double* J, double* H /*6x3*/) {
    const double wO = J[18];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) H[3 * i + j] = (J[6 + i] * J[j] + J[12 + i] * J[3 + j]) * wO;
}

// The Schur operand of one point.  With M = Hll + lambda I = C C^T (Cholesky, C lower) and Ci = C^-1:
//     Hpl M^-1 Hpl^T = (Hpl Ci^T)(Hpl Ci^T)^T     and     Hpl M^-1 bl = (Hpl Ci^T)(Ci bl),
// so ONE operand G = Hpl Ci^T per edge block (and Ci bl in the right-hand side's row) gives the whole product as G G^T: half the
// operand stores of the point pass and half the slab loads of the product that the pair (Hpl M^-1, Hpl) took.  (M is positive
// definite for lambda > 0; g2o forms Hpl (Hll^-1 Hpl^T) with the explicit inverse, block_solver.hpp:381-432.)
struct PointChol { double i00, i10, i11, i20, i21, i22; };
__device__ __forceinline__ PointChol point_chol(double a, double b, double c, double d, double e, double f) {   // M = [a b c; b d e; c e f]
    PointChol q;
    const double c00 = sqrt(a);
    q.i00 = 1.0 / c00;
    const double c10 = b * q.i00, c20 = c * q.i00;
    const double c11 = sqrt(d - c10 * c10);
    q.i11 = 1.0 / c11;
    const double c21 = (e - c20 * c10) * q.i11;
    const double c22 = sqrt(f - c20 * c20 - c21 * c21);
    q.i22 = 1.0 / c22;
    q.i10 = -c10 * q.i00 * q.i11;
    q.i21 = -c21 * q.i11 * q.i22;
    q.i20 = -(c20 * q.i00 + c21 * q.i10) * q.i22;
    return q;
}

__global__ __launch_bounds__(256) void k_prepare(BaWin* wins) {   // BA_PG lanes per point, like k_point_reduce
    const BaWin& W = wins[blockIdx.y];
    BaState* st = BA_ST(wins, blockIdx.y);
    if (st->done) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int p = t / BA_PG, g = t % BA_PG;
    if (p >= W.n_pt) return;
    const double lambda = st->lambda;
    const double* h = W.Hll + 6 * (size_t)p;
    const double a = h[0] + lambda, b = h[1], c = h[2], d = h[3] + lambda, e_ = h[4], f = h[5] + lambda;
    // symmetric 3x3 inverse by cofactors (every lane of the group computes it: cheaper than a broadcast)
    const double c0 = d * f - e_ * e_, c1 = e_ * c - b * f, c2 = b * e_ - d * c;
    const double det = a * c0 + b * c1 + c * c2;
    const double id = 1.0 / det;
    double Di[6];
    Di[0] = c0 * id; Di[1] = c1 * id; Di[2] = c2 * id;
    Di[3] = (a * f - c * c) * id; Di[4] = (b * c - a * e_) * id; Di[5] = (a * d - b * b) * id;
    const size_t K = (size_t)W.Kpad;
    const PointChol L = point_chol(a, b, c, d, e_, f);
    if (g == 0) {
        for (int i = 0; i < 6; ++i) W.Dinv[6 * (size_t)p + i] = Di[i];
        const double* bl = W.bl + 3 * (size_t)p;
        gdouble* q = (gdouble*)W.GA + (size_t)W.nS * K + 3 * (size_t)p;
        q[0] = L.i00 * bl[0]; q[1] = L.i10 * bl[0] + L.i11 * bl[1]; q[2] = L.i20 * bl[0] + L.i21 * bl[1] + L.i22 * bl[2];
    }
    for (int i = W.pt_ptr[p] + g; i < W.pt_ptr[p + 1]; i += BA_PG) {
        const int e = W.pt_edges[i];
        if (!W.e_active[e]) continue;
        const int col = W.pose_col[W.e_kf[e]];
        if (col < 0) continue;
        double H[18];
        if (W.nrow == 3) hpl_of3(W.e_jac + BA_JAC_STEREO * (size_t)e, H);
        else hpl_of(W.e_jac + 21 * (size_t)e, H);
        for (int r = 0; r < 6; ++r) {
            const double h0 = H[3 * r], h1 = H[3 * r + 1], h2 = H[3 * r + 2];
            const size_t o = (size_t)(6 * col + r) * K + 3 * (size_t)p;
            W.GA[o] = h0 * L.i00;
            W.GA[o + 1] = h0 * L.i10 + h1 * L.i11;
            W.GA[o + 2] = h0 * L.i20 + h1 * L.i21 + h2 * L.i22;
        }
    }
}

// ---- S1 + S2 + S5 in one launch for every slot after a stage's first: only the first trial of a stage takes its lambda
// from the maximum over ALL diagonal blocks (computeLambdaInit) and therefore needs a kernel boundary between the
// reductions and the damping; from then on lambda is in the state when the slot starts.  The lane that k_point_reduce
// and k_prepare give an edge to computes that edge's Jacobian itself (and stores it for k_pose_reduce and
// k_backsub_update); after a rejected trial only the damping part runs.  Same expressions, same summation order.
__global__ __launch_bounds__(256) void k_point_pass(BaWin* wins) {
    const BaWin& W = wins[blockIdx.y];
    BaState* st = BA_ST(wins, blockIdx.y);
    if (st->done) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int p = t / BA_PG, g = t % BA_PG;
    const bool live = p < W.n_pt;
    const bool lin = st->need_linearize != 0;
    double h[6] = {0, 0, 0, 0, 0, 0}, b[3] = {0, 0, 0};
    if (lin) {
        const bool robust = st->robust != 0;
        if (live)
            for (int i = W.pt_ptr[p] + g; i < W.pt_ptr[p + 1]; i += BA_PG) {
                const int e = W.pt_edges[i];
                if (!W.e_active[e]) continue;
                if (W.nrow == 3) {
                    double J3[BA_JAC_STEREO];
                    edge_jacobian3(W, robust, e, J3);
                    double* J3m = W.e_jac + BA_JAC_STEREO * (size_t)e;
#pragma unroll
                    for (int k = 0; k < BA_JAC_STEREO; ++k) J3m[k] = J3[k];
                    point_accum3(J3, h, b);
                    continue;
                }
                double J[21];
                edge_jacobian(W, robust, e, J);
                double* Jm = W.e_jac + 21 * (size_t)e;
#pragma unroll
                for (int k = 0; k < 21; ++k) Jm[k] = J[k];
                const double wO = J[18], r0 = J[19], r1 = J[20];
                b[0] += J[0] * r0 + J[3] * r1; b[1] += J[1] * r0 + J[4] * r1; b[2] += J[2] * r0 + J[5] * r1;
                h[0] += (J[0] * J[0] + J[3] * J[3]) * wO; h[1] += (J[0] * J[1] + J[3] * J[4]) * wO; h[2] += (J[0] * J[2] + J[3] * J[5]) * wO;
                h[3] += (J[1] * J[1] + J[4] * J[4]) * wO; h[4] += (J[1] * J[2] + J[4] * J[5]) * wO; h[5] += (J[2] * J[2] + J[5] * J[5]) * wO;
            }
#pragma unroll
        for (int i = 0; i < 6; ++i) h[i] = group_sum(h[i]);
#pragma unroll
        for (int i = 0; i < 3; ++i) b[i] = group_sum(b[i]);
        if (live && g == 0) {
            for (int i = 0; i < 6; ++i) W.Hll[6 * (size_t)p + i] = h[i];
            for (int i = 0; i < 3; ++i) W.bl[3 * (size_t)p + i] = b[i];
        }
    } else if (live) {
        for (int i = 0; i < 6; ++i) h[i] = W.Hll[6 * (size_t)p + i];
        for (int i = 0; i < 3; ++i) b[i] = W.bl[3 * (size_t)p + i];
    }
    if (!live) return;
    const double lambda = st->lambda;
    const double a = h[0] + lambda, bq = h[1], c = h[2], d = h[3] + lambda, e_ = h[4], f = h[5] + lambda;
    const double c0 = d * f - e_ * e_, c1 = e_ * c - bq * f, c2 = bq * e_ - d * c;
    const double det = a * c0 + bq * c1 + c * c2;
    const double id = 1.0 / det;
    double Di[6];
    Di[0] = c0 * id; Di[1] = c1 * id; Di[2] = c2 * id;
    Di[3] = (a * f - c * c) * id; Di[4] = (bq * c - a * e_) * id; Di[5] = (a * d - bq * bq) * id;
    const size_t K = (size_t)W.Kpad;
    const PointChol L = point_chol(a, bq, c, d, e_, f);
    if (g == 0) {
        for (int i = 0; i < 6; ++i) W.Dinv[6 * (size_t)p + i] = Di[i];
        gdouble* q = (gdouble*)W.GA + (size_t)W.nS * K + 3 * (size_t)p;
        q[0] = L.i00 * b[0]; q[1] = L.i10 * b[0] + L.i11 * b[1]; q[2] = L.i20 * b[0] + L.i21 * b[1] + L.i22 * b[2];
    }
    for (int i = W.pt_ptr[p] + g; i < W.pt_ptr[p + 1]; i += BA_PG) {
        const int e = W.pt_edges[i];
        if (!W.e_active[e]) continue;
        const int col = W.pose_col[W.e_kf[e]];
        if (col < 0) continue;
        double H[18];
        if (W.nrow == 3) hpl_of3(W.e_jac + BA_JAC_STEREO * (size_t)e, H);
        else hpl_of(W.e_jac + 21 * (size_t)e, H);   // this lane's own store when lin (same address, same thread)
        for (int r = 0; r < 6; ++r) {
            const double h0 = H[3 * r], h1 = H[3 * r + 1], h2 = H[3 * r + 2];
            const size_t o = (size_t)(6 * col + r) * K + 3 * (size_t)p;
            W.GA[o] = h0 * L.i00;
            W.GA[o + 1] = h0 * L.i10 + h1 * L.i11;
            W.GA[o + 2] = h0 * L.i20 + h1 * L.i21 + h2 * L.i22;
        }
    }
}

// ---- S6: split-K dense product  part[s] = GA[:, ks] * GA[:, ks]^T  on fp64 MFMA (GA = Hpl Ci^T: point_chol) ------------------
// Workgroup = 4 wavefronts, one 64x64 macro tile (I <= J) of one k split.  Wave w owns rows
// 16w..16w+15 and all four 16-wide column tiles (4 x double4 accumulators).  Slabs of 64 rows x
// 32 k of both operands are staged in LDS (pitch 34 doubles: conflict-free ds_read_b64 for the
// MFMA operand pattern lane -> [row = lane&15][k = lane>>4]).
#define LDS_PITCH 34

// does the reduced system need tile pair (I, J), I <= J, of the product?  Not when the two row tiles share no k range (the block
// is zero: k_schur_reduce writes the zeros itself), and not when the banded solver takes the window and the whole tile lies
// outside the band (nobody reads it) -- except for the tile column that holds the right-hand side (row nS of the operand)
__device__ __forceinline__ bool schur_tile_needed(const BaWin& W, int I, int J) {
    if (max(W.tile_alo[I], W.tile_blo[J]) >= min(W.tile_ahi[I], W.tile_bhi[J])) return false;
    if (W.solver == BA_SOLVER_BAND && BA_TILE * J - (BA_TILE * I + BA_TILE - 1) > W.band && W.nS / BA_TILE != J) return false;
    return true;
}

__device__ __forceinline__ void schur_body(const BaWin& W, const BaState* st, int tile, int s, int nsplit) {
    if (st->done) return;
    const int T = W.Npad / BA_TILE;
    // What the workgroup multiplies: rows rbaseA + r of GA against its rows rbaseB + r (r = 0 .. 63; rows past the matrix are read as its
    // last row: a row / column of the result nobody reads --, row 63 of the B operand is row lastB) over the k range
    // [k0, kend), into the 64 x 64 tile at `out` (row pitch opitch).
    int rbaseA, rbaseB, lastB, k0, kend, opitch;
    gdouble* out;
    if (W.sf_groups) {
        // floating row windows (BaWin::sf_*): group g = the launch's workgroup number inside the window
        const int g = tile * nsplit + s;
        if (g >= W.sf_groups) return;
        rbaseA = rbaseB = W.sf_row[g]; lastB = W.nS;
        k0 = W.sf_k0[g] * BA_KC; kend = W.sf_k1[g] * BA_KC;
        out = (gdouble*)W.part + (size_t)g * (BA_TILE * BA_TILE); opitch = BA_TILE;
    } else {
        // `tile` enumerates upper-triangular macro tiles
        int I = 0, rem = tile;
        while (I < T && rem >= T - I) { rem -= T - I; ++I; }
        const int J = I + rem;
        if (I >= T || !schur_tile_needed(W, I, J)) return;
        // Only the k range in which BOTH row tiles have non-zeros is multiplied (points are sorted by their first observing
        // keyframe, ba_api.hip): its slabs of BA_KC are dealt to the launch's nsplit <= BA_SPLITS splits (gridDim.y: sixteen for a single
        // window, which needs the parallelism; eight for a batch, whose windows already fill the chip -- half of the partial
        // tiles to write and to sum); a split without a slab stores zeros.
        const int klo = max(W.tile_alo[I], W.tile_blo[J]), khi = min(W.tile_ahi[I], W.tile_bhi[J]);
        const int nslab = khi > klo ? (khi - klo) / BA_KC : 0;
        k0 = klo + (int)((long)nslab * s / nsplit) * BA_KC; kend = klo + (int)((long)nslab * (s + 1) / nsplit) * BA_KC;
        rbaseA = I * BA_TILE; rbaseB = J * BA_TILE; lastB = J * BA_TILE + BA_TILE - 1;
        out = (gdouble*)W.part + (size_t)s * W.Npad * W.Npad + (size_t)(I * BA_TILE) * W.Npad + J * BA_TILE; opitch = W.Npad;
    }
    __shared__ __attribute__((aligned(16))) double As[BA_TILE * LDS_PITCH];
    __shared__ __attribute__((aligned(16))) double Bs[BA_TILE * LDS_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t K = (size_t)W.Kpad;
    double4_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (double4_t){0, 0, 0, 0};
    // software pipeline: the next slab is fetched into registers while the current one feeds the MFMAs; 16 bytes per lane and load
    // (a slab row is 32 doubles = 16 lanes; k ranges are multiples of BA_KC, Kpad of BA_KC * BA_SPLITS, so every address is 16-byte aligned)
    typedef __attribute__((address_space(1))) double2_t gdouble2s;
    const gdouble2s* srcA[4];
    const gdouble2s* srcB[4];
    {
        const int lastrow = W.Npad - 1, c = (tid & 15) * 2;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = (tid >> 4) + 16 * u;
            srcA[u] = (const gdouble2s*)((const gdouble*)W.GA + (size_t)min(rbaseA + r, lastrow) * K + c);
            srcB[u] = (const gdouble2s*)((const gdouble*)W.GA + (size_t)(r == BA_TILE - 1 ? lastB : min(rbaseB + r, lastrow)) * K + c);
        }
    }
    // A floating window's B rows are its A rows but for the last: its lanes copy three of their four B operands from the A loads and
    // fetch the fourth (row 63's lanes: the right-hand side's row; the others their A row once more, a cache hit) -- every load of
    // either form is unconditional, a load behind a branch would be waited for on its own.
    double2_t pa[4], pb[4];
    auto product = [&](auto sf_c) {
        constexpr bool SF = decltype(sf_c)::value;
        auto fetch = [&](int k) {
#pragma unroll
            for (int u = 0; u < 4; ++u) pa[u] = srcA[u][k >> 1];
            if (SF) { pb[0] = pa[0]; pb[1] = pa[1]; pb[2] = pa[2]; pb[3] = srcB[3][k >> 1]; }
            else {
#pragma unroll
                for (int u = 0; u < 4; ++u) pb[u] = srcB[u][k >> 1];
            }
        };
        if (kend > k0) fetch(k0);
        for (int kk = k0; kk < kend; kk += BA_KC) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = tid + 256 * u, r = i >> 4, c = (i & 15) * 2;
                *reinterpret_cast<double2_t*>(&As[r * LDS_PITCH + c]) = pa[u];
                *reinterpret_cast<double2_t*>(&Bs[r * LDS_PITCH + c]) = pb[u];
            }
            __syncthreads();
            if (kk + BA_KC < kend) fetch(kk + BA_KC);
#pragma unroll
            for (int ks = 0; ks < BA_KC; ks += 4) {
                const double a = As[(16 * wv + (lane & 15)) * LDS_PITCH + ks + (lane >> 4)];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const double b = Bs[(16 * j + (lane & 15)) * LDS_PITCH + ks + (lane >> 4)];
                    acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
                }
            }
        }
    };
    if (W.sf_groups) product(std::true_type{}); else product(std::false_type{});
    // C/D layout of v_mfma_f64_16x16x4: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            out[(size_t)(16 * wv + (lane >> 4) + 4 * r) * opitch + 16 * j + (lane & 15)] = acc[j][r];
}

__global__ __launch_bounds__(256) void k_schur(BaWin* wins) { schur_body(wins[blockIdx.z], BA_ST(wins, blockIdx.z), blockIdx.x, blockIdx.y, gridDim.y); }

// Both in one launch for the slots after a stage's first (neither needs the other; k_schur_reduce needs both): the pose blocks ride
// as extra workgroups behind the Schur tiles -- keyframe (x - ntiles) * (splits of the launch) + y.
__global__ __launch_bounds__(256) void k_schur_pose(BaWin* wins, int ntiles) {
    const BaWin& W = wins[blockIdx.z];
    if ((int)blockIdx.x < ntiles) schur_body(W, BA_ST(wins, blockIdx.z), blockIdx.x, blockIdx.y, gridDim.y);
    else pose_reduce_body(W, BA_ST(wins, blockIdx.z), ((int)blockIdx.x - ntiles) * (int)gridDim.y + (int)blockIdx.y);
}

__host__ __device__ inline int ldlt_band_rs(int bw);
__host__ __device__ inline bool ldlt_band_ok(int n, int bw);

// ---- S7: S = Hpp + lambda*I - sum_s part[s],  b_s = bp - coeff --------------------------------------
// A window of the blocked solver: thread -> entry (r, c) of the upper triangle, the full matrix is written (both halves).  A banded
// window: ONE WAVE PER ROW -- lanes 0 .. band are the row's entries r .. r + band, lane 63 its right-hand side (band <= 59) -- and only the
// solver's LDS image is written: as thread -> entry, four waves of five held no entry of the band (one launch of a batch of 64: 62 us).
// The partial sums are added in split order whatever the mapping.
__global__ __launch_bounds__(256) void k_schur_reduce(BaWin* wins, int nsplit) {
    const BaWin& W = wins[blockIdx.y];
    BaState* st = BA_ST(wins, blockIdx.y);
    if (st->done) return;
    const int n = W.nS, N = W.Npad;
    const bool banded = W.solver == BA_SOLVER_BAND;
    int r, c;
    if (banded) {
        const int lane = threadIdx.x & 63;
        r = blockIdx.x * 4 + (threadIdx.x >> 6);
        c = lane == 63 ? n : r + lane;
        if (r >= n || (lane != 63 && (lane > W.band || c >= n))) return;
    } else {
        const int idx = blockIdx.x * 256 + threadIdx.x;
        r = idx / N; c = idx - r * N;
        if (r >= n || c > n || c < r) return;
    }
    double v = 0;
    if (W.sf_groups) {
        // floating windows: the tiles of the groups that hold both rows (the right-hand side: row r), in group order
        const int g0 = c == n ? W.sf_glo[r] : W.sf_glo[c], g1 = W.sf_ghi[r];
        const gdouble* part = (const gdouble*)W.part;
        for (int q = g0; q <= g1; q += 4) {   // four loads in flight; a group behind the last adds 0
            double x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = min(q + u, g1), lo = W.sf_row[g];
                x[u] = part[(size_t)g * (BA_TILE * BA_TILE) + (r - lo) * BA_TILE + (c == n ? BA_TILE - 1 : c - lo)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) v += q + u <= g1 ? x[u] : 0.0;
        }
    } else if (schur_tile_needed(W, r / BA_TILE, c / BA_TILE)) {   // (else the product's block is zero and no workgroup wrote its partials)
        const gdouble* p = (const gdouble*)W.part + (size_t)r * N + c;
        const size_t NN = (size_t)N * N;
        for (int s0 = 0; s0 < nsplit; s0 += 4) {   // four loads in flight; a split behind the last adds 0
            double x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = p[(size_t)min(s0 + u, nsplit - 1) * NN];
#pragma unroll
            for (int u = 0; u < 4; ++u) v += s0 + u < nsplit ? x[u] : 0.0;
        }
    }
    if (c == n) {
        W.rhs[r] = W.bp[r] - v;  // _bschur = _b - coefficients
        return;
    }
    double hv = 0;
    if (r / 6 == c / 6) {
        hv = W.Hpp[36 * (size_t)(r / 6) + 6 * (r % 6) + (c % 6)];
        if (r == c) hv += st->lambda;
    }
    const double sv = hv - v;
    if (banded) {   // entry (row c, column r) of the lower band in ldlt_band_solve's layout; nobody reads S
        const int bw = W.band;
        W.Sb[(size_t)c * ldlt_band_rs(bw) + (r - c + bw + 3)] = sv;
    } else {
        W.S[(size_t)r * N + c] = sv;
        W.S[(size_t)c * N + r] = sv;
    }
}

// ---- S8: dense blocked LDLt (no pivoting) of the reduced system + solve, one workgroup -------------
// In place on the lower triangle of S (row-major, pitch Npad).  The right-hand side rides along as
// row n of the matrix, so the forward substitution is part of the factorisation: after the last
// panel, row n holds y = D^-1 L^-1 b; a blocked backward substitution with L^T finishes.
// Per 32-column panel: (1) diagonal block + panel rows are staged in LDS with coalesced loads,
// (2) wavefront 0 factors the diagonal block (one (r,c) pair per lane per step, pair table in LDS),
// (3) every row below solves its 32 entries out of LDS, (4) the trailing matrix takes the rank-32
// update A22 -= (L21 D) D^-1 (L21 D)^T on v_mfma_f64_16x16x4_f64, 16x16 tiles spread over 8 waves.
// A pivot that is exactly zero fails the solve like SimplicialLDLT's NumericalIssue.
#define LD_NB 32
#define LD_P (LD_NB + 1)
#define LD_THREADS 512
#ifndef LD_CDEPTH
#define LD_CDEPTH 4         // C tiles of the trailing update in flight per wave
#endif
#define LD_PAIRS (LD_NB * (LD_NB - 1) / 2)

// Factor one 32x32 diagonal block in LDS (unit L below, D on the diagonal) with the whole
// workgroup: at step k thread p updates one (r, c) pair of the trailing triangle,
// a[r][c] -= a[r][k] a[c][k] / d_k, one barrier per step.  (Register-resident variants of this
// serial section — row per lane with v_readlane or ds_bpermute broadcasts — made hipcc spill
// SGPRs/VGPRs to scratch, ~1 us per reload; the LDS form is the robust one.)
__device__ __forceinline__ double fast_recip(double d) {
    double x = __builtin_amdgcn_rcp(d);   // v_rcp_f64
    x = x * (2.0 - d * x);                // one Newton step: ~1e-16 relative, no v_div_* sequence
    return x;
}

__device__ __forceinline__ void ldlt_factor_diag(double* Dg, double* s_invd, double* s_corr, double* s_dval,
                                                 int nb, int tid, int* s_fail) {
    // TWO columns per step (16 barriers per block instead of 32): with d0 = a[k][k], f = a[k+1][k] / d0 and
    // d1 = a[k+1][k+1] - a[k+1][k] f, every pair (r, c), c >= k+2, takes the rank-2 update
    //     a[r][c] -= a[r][k] a[c][k] / d0 + a'[r] a'[c] / d1,   a'[x] = a[x][k+1] - a[x][k] f
    // Columns k and k+1 themselves are never written inside the loop (their final L*d values are a[.][k] and a'[.]);
    // the unit-L entries are produced at the end from the untouched columns.
    // Every thread OWNS one entry of the lower triangle (528 entries: threads 0..15 own a second one) and keeps it in a
    // register through all the steps; it only stores it to LDS when its column is about to become a pivot column.  A
    // step is then ONE round of LDS reads (three pivot entries + four column entries, all issued right after the
    // barrier), the pivot arithmetic, six FMAs and at most one store -- no pair table, no read-modify-write.
    int r0, c0, r1 = 0, c1 = 0;
    {
        int q = tid;
        r0 = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
        while ((r0 + 1) * (r0 + 2) / 2 <= q) ++r0;
        while (r0 * (r0 + 1) / 2 > q) --r0;
        c0 = q - r0 * (r0 + 1) / 2;
        if (tid < LD_NB * (LD_NB + 1) / 2 - LD_THREADS) {   // 16 leftover entries: the tail of row 31
            q = LD_THREADS + tid;
            r1 = LD_NB - 1; c1 = q - r1 * (r1 + 1) / 2;
        }
    }
    const bool two_entries = tid < LD_NB * (LD_NB + 1) / 2 - LD_THREADS;
    double v0 = Dg[r0 * LD_P + c0], v1 = two_entries ? Dg[r1 * LD_P + c1] : 0.0;
    // The step is bound by the instructions every wave issues (two waves per SIMD run the same stream), not by latency:
    // every load is unconditional (rows and columns < 32 are always inside the block; an inactive entry computes on
    // whatever it reads and keeps its old value through a select), the second entry of threads 0..15 sits behind a
    // wave-uniform branch, and 1/d1 comes from the 2x2 determinant so that both reciprocals start at once:
    //     1/d1 = d0 / (d0 d11 - l10^2).
    const double* pr0 = Dg + r0 * LD_P;
    const double* pc0 = Dg + c0 * LD_P;
    const double* pr1 = Dg + r1 * LD_P;
    const double* pc1 = Dg + c1 * LD_P;
    const bool in0 = r0 < nb, in1 = two_entries && r1 < nb;
    for (int k = 0; k < nb; k += 2) {
        const bool two = k + 1 < nb;   // (row / column nb of a short block hold zeros)
        const double d0 = Dg[k * LD_P + k], l10 = Dg[(k + 1) * LD_P + k], d11 = Dg[(k + 1) * LD_P + k + 1];
        const double ark0 = pr0[k], ar10 = pr0[k + 1], ack0 = pc0[k], ac10 = pc0[k + 1];
        const bool bad0 = (d0 == 0.0 || !(fabs(d0) <= DBL_MAX));
        const double det = d0 * d11 - l10 * l10;
        const double inv0 = bad0 ? 0.0 : fast_recip(d0);
        const bool bad1 = two && (det == 0.0 || !(fabs(det) <= DBL_MAX));
        const double inv1 = (two && !bad1 && !bad0) ? d0 * fast_recip(det) : 0.0;
        const double f = l10 * inv0;
        {
            const double ar1 = ar10 - ark0 * f, ac1 = ac10 - ack0 * f;
            const double nv = v0 - (ark0 * ack0 * inv0 + ar1 * ac1 * inv1);
            const bool act0 = in0 && c0 >= k + 2;
            v0 = act0 ? nv : v0;
            if (act0 && c0 <= k + 3) Dg[r0 * LD_P + c0] = v0;   // column c0 is a pivot column of the next step
        }
        if (tid < 64) {   // wave-uniform: the 16 leftover entries (tail of row 31) live in wave 0
            const double ark1 = pr1[k], ar11 = pr1[k + 1], ack1 = pc1[k], ac11 = pc1[k + 1];
            const double ar1 = ar11 - ark1 * f, ac1 = ac11 - ack1 * f;
            const double nv = v1 - (ark1 * ack1 * inv0 + ar1 * ac1 * inv1);
            const bool act1 = in1 && c1 >= k + 2;
            v1 = act1 ? nv : v1;
            if (act1 && c1 <= k + 3) Dg[r1 * LD_P + c1] = v1;
        }
        if (tid == 0) {
            s_invd[k] = inv0;
            if (two) { s_invd[k + 1] = inv1; s_corr[k + 1] = f; s_dval[k + 1] = d11 - l10 * f; }
            if (bad0 || bad1) *s_fail = 1;
        }
        __syncthreads();
    }
    for (int k = nb + tid; k < LD_NB; k += LD_THREADS) s_invd[k] = 0.0;
    __syncthreads();
    // unit-L entries and the diagonal of the odd columns: thread -> row r, column pair (c, c+1), c even
    for (int i = tid; i < LD_NB * (LD_NB / 2); i += LD_THREADS) {
        const int r = i >> 4, c = (i & 15) * 2;
        if (r >= nb) continue;
        const double a = Dg[r * LD_P + c], b2 = Dg[r * LD_P + c + 1];
        if (r > c) Dg[r * LD_P + c] = a * s_invd[c];
        if (c + 1 < nb) {
            if (r > c + 1) Dg[r * LD_P + c + 1] = (b2 - a * s_corr[c + 1]) * s_invd[c + 1];
            else if (r == c + 1) Dg[r * LD_P + r] = s_dval[c + 1];
        }
    }
}

// rows below (and the rhs row): w = a L11^-T (w = L*d), row in registers.  RIGHT-looking: once w_k is final, the
// 31-k updates w_m -= w_k L11[m][k] are independent of each other, so the dependent chain is 32 long instead of the
// 496 of the dot-product form (which ran at one FMA + LDS read per ~45 cycles).
__device__ __forceinline__ void ldlt_rows(double* Wd, const double* Dg, int rows, int tid) {
    for (int r = tid; r < rows; r += LD_THREADS) {
        double* wrow = Wd + r * LD_P;
        double w[LD_NB];
#pragma unroll
        for (int k = 0; k < LD_NB; ++k) w[k] = wrow[k];
        // software-pipelined: an entry of column k+1 of L11 is requested from LDS into the register its column-k entry has just
        // left (a second set of 32 registers for the next column made the kernel spill 36 VGPRs)
        double lc[LD_NB];
#pragma unroll
        for (int m = 1; m < LD_NB; ++m) lc[m] = Dg[m * LD_P];
#pragma unroll
        for (int k = 0; k < LD_NB - 1; ++k) {
            const double wk = w[k];
#pragma unroll
            for (int m = k + 1; m < LD_NB; ++m) {
                w[m] -= wk * lc[m];
                if (m >= k + 2) lc[m] = Dg[m * LD_P + k + 1];
            }
            __builtin_amdgcn_sched_barrier(0);  // keeps the reads of later columns from being hoisted (spills)
        }
#pragma unroll
        for (int k = 1; k < LD_NB; ++k) wrow[k] = w[k];
    }
}

// back-substitution inside one 32-row block: lane k owns x_k, the columns of L11 sit in registers
__device__ __forceinline__ void ldlt_back_block(const gdouble* S, int N, double* xs, int jb, int nb, int lane) {
    const int k = lane;
    double v = (k < nb) ? xs[jb + k] : 0.0;
    double col[LD_NB];
#pragma unroll
    for (int m = 0; m < LD_NB; ++m) col[m] = (k < nb && m < nb && m > k) ? S[(size_t)(jb + m) * N + jb + k] : 0.0;
#pragma unroll
    for (int m = LD_NB - 1; m >= 0; --m) {
        const double xm = readlane_d(v, m);   // final once every higher index has been applied
        if (m < nb && k < m) v -= col[m] * xm;
    }
    if (k < nb) xs[jb + k] = v;
}

// ---- banded path ---------------------------------------------------------------------------------------------
// A local window whose keyframes share points only with their neighbours (ORB-SLAM's local BA: co-visibility falls off
// with the distance along the trajectory) gives a reduced system with a narrow row envelope, and LDLt without pivoting
// never fills outside it.  With half bandwidth bw the lower band (n rows of bw + 4 doubles: columns r - bw - 3 ..
// r - bw - 1, always zero -- a four-column read may begin up to three columns left of the band --, then r - bw .. r)
// fits the CU's LDS for n = 300, bw <= 59, and the whole solve runs there: no panel staging from L2, no write-back, no
// trailing update through L2.  Entry (r, c) of the band lives at Ab[r * RS + c - r + bw + 3].
//
// The factorisation is a right-looking BLOCK LDLt with 4 x 4 pivot blocks, n / 4 steps, one barrier per step.  With
// E the pivot block of step k as it stands after the earlier steps, R the rows below it in the pivot's four columns
// ("raw panel": also final after the earlier steps) and G = E^-1, the trailing matrix takes
//     A(r, c) -= R_r G R_c^T,
// a rank-4 update -- exactly one v_mfma_f64_16x16x4_f64 per 16 x 16 tile: A operand R_r G (four raw entries and one
// row of G per lane), B operand the raw entry itself.  The window the update can reach (rows / columns k + 4 ..
// k + 3 + bw) lies inside a 5 x 5 triangle of tiles; the tiles LIVE IN ACCUMULATORS for as long as they are in the
// window (wave I mod 5 owns block row I: its tiles share one A operand) and are read from the band once, when their
// block row enters.  What goes back to LDS per step is
// only the next raw panel (4 columns) and the next-but-one pivot block.
//   waves 0..4  tiles: operands, MFMA, the next raw panel (columns k + 4 .. k + 7) and a preview of the block
//               (k + 8 .. k + 11)^2 for the look-ahead;
//   wave 5      PIVOT LOOK-AHEAD: the next block E' = E_preview - P G P^T (P = its rows of the current raw panel), its
//               LDLt, G' = E'^-1 for the next step's operands and T' = L4^-T D4^-1 / L4 for the unit-L form;
//   wave 6      the right-hand side ("row n"): y_c -= R_c G y_k;
//   wave 7      only meets the barriers.
// Afterwards the raw panels become the scalar unit-L factor (L(r, k..k+3) = R_r T, L4 inside a block), z = T^T y, and
// the blocked backward substitution runs out of LDS as before.
#ifndef LB_PIV_WAVE
#define LB_PIV_WAVE 3
#define LB_RHS_WAVE 7
#endif
#define LD_BAND_LDS (150 * 1024)     // dynamic LDS the kernel may use (bak_ldlt_smem requests at least this much when it fits)
// Row stride EVEN: a column of the band, A(c, k) for c = k + 1, k + 2, .., is a walk of RS - 1 doubles per row, and an odd number
// of doubles per step spreads 32 lanes over 32 different bank pairs (with RS = 49 they all fell on two).
__host__ __device__ inline int ldlt_band_rs(int bw) { return (bw + 5) & ~1; }
__host__ __device__ inline int ldlt_band_ylen(int n) { return (n + 9) & ~1; }
__host__ __device__ inline size_t ldlt_band_bytes(int n, int bw) {
    return sizeof(double) * ((size_t)(n + 1) * ldlt_band_rs(bw) + ldlt_band_ylen(n) + 16 * (size_t)((n + 3) / 4 + 1));
}
// bw <= 59: the reachable rows k + 4 .. k + 3 + bw stay within five block rows of the pivot's block column
__host__ __device__ inline bool ldlt_band_ok(int n, int bw) { return n > 0 && bw >= 8 && bw <= 59 && ldlt_band_bytes(n, bw) <= LD_BAND_LDS; }

// LDLt of a symmetric 4 x 4 block (lower entries e: 00 10 11 20 21 22 30 31 32 33; rows >= nv are padding and count as
// identity) and its NEGATED inverse G (row-major, symmetric: every user subtracts).
__device__ __forceinline__ void ldlt_piv4(const double* e, int nv, double* G, bool& bad) {
    double e00 = e[0], e10 = e[1], e11 = e[2], e20 = e[3], e21 = e[4], e22 = e[5], e30 = e[6], e31 = e[7], e32 = e[8], e33 = e[9];
    if (nv < 4) { e30 = 0; e31 = 0; e32 = 0; e33 = 1; }
    if (nv < 3) { e20 = 0; e21 = 0; e22 = 1; }
    if (nv < 2) { e10 = 0; e11 = 1; }
    // The four pivots are a dependent chain (this wave's step is the solve's critical path): every pivot is formed as
    // d' = d - (a a x) w with x = v_rcp_f64(d_prev), w = 2 - d_prev x its Newton factor, so that only rcp -> {w, a a x} -> fma
    // lie between two reciprocals; everything else (the multipliers l, the other entries) hangs off the side.
    const double x0 = __builtin_amdgcn_rcp(e00), w0 = 2.0 - e00 * x0, i0 = x0 * w0;
    const double d1 = e11 - ((e10 * e10) * x0) * w0;
    const double l10 = e10 * i0, l20 = e20 * i0, l30 = e30 * i0;
    e21 -= l20 * e10; e31 -= l30 * e10; e22 -= l20 * e20; e32 -= l30 * e20; e33 -= l30 * e30;
    const double x1 = __builtin_amdgcn_rcp(d1), w1 = 2.0 - d1 * x1, i1 = x1 * w1;
    const double d2 = e22 - ((e21 * e21) * x1) * w1;
    const double l21 = e21 * i1, l31 = e31 * i1;
    e32 -= l31 * e21; e33 -= l31 * e31;
    const double x2 = __builtin_amdgcn_rcp(d2), w2 = 2.0 - d2 * x2, i2 = x2 * w2;
    const double d3 = e33 - ((e32 * e32) * x2) * w2;
    const double l32 = e32 * i2;
    const double x3 = __builtin_amdgcn_rcp(d3), i3 = x3 * (2.0 - d3 * x3);
    auto isbad = [](double d) { return d == 0.0 || !(fabs(d) <= DBL_MAX); };
    bad = isbad(e00) || isbad(d1) || isbad(d2) || isbad(d3);
    // M = L^-1 (unit lower)
    const double m10 = -l10, m21 = -l21, m32 = -l32;
    const double m20 = -(l20 + l21 * m10), m31 = -(l31 + l32 * m21);
    const double m30 = -(l30 + l31 * m10 + l32 * m20);
    const double t01 = m10 * i1, t02 = m20 * i2, t12 = m21 * i2, t03 = m30 * i3, t13 = m31 * i3, t23 = m32 * i3;
    const double g33 = i3, g23 = t23, g13 = t13, g03 = t03;
    const double g22 = i2 + t23 * m32, g12 = t12 + t13 * m32, g02 = t02 + t03 * m32;
    const double g11 = (i1 + t12 * m21) + t13 * m31, g01 = (t01 + t02 * m21) + t03 * m31;
    const double g00 = (i0 + t01 * m10) + (t02 * m20 + t03 * m30);
    G[0] = -g00; G[1] = -g01; G[2] = -g02; G[3] = -g03;
    G[4] = -g01; G[5] = -g11; G[6] = -g12; G[7] = -g13;
    G[8] = -g02; G[9] = -g12; G[10] = -g22; G[11] = -g23;
    G[12] = -g03; G[13] = -g13; G[14] = -g23; G[15] = -g33;
}


__device__ __forceinline__ void ldlt_band_solve(const BaWin& W, BaState* st, double* sm, int* s_fail) {
    const int n = W.nS, N = W.Npad, bw = W.band, RS = ldlt_band_rs(bw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // roles: waves 0, 1, 2, 4, 5 update block rows (uw mod 5), wave 6 helps them with the window's lower left corner, wave 3 is the
    // pivot wave, wave 7 the right-hand side: with waves dealt round-robin to the four SIMDs the pivot chain then shares its SIMD
    // only with the rhs wave, not with fp64 MFMAs
    const int uw = wv < 3 ? wv : (wv == 4 || wv == 5) ? wv - 1 : -1;
    const bool is_piv = wv == LB_PIV_WAVE, is_rhs = wv == LB_RHS_WAVE, is_hlp = wv == 6;
    double* Ab = sm;                          // (n + 1) x RS, the extra row is zero
    double* y = Ab + (size_t)(n + 1) * RS;    // right-hand side -> block forward substitution -> z -> x (zeros behind n)
    double* fac = y + ldlt_band_ylen(n);      // 16 doubles per pivot block: -E^-1 (ldlt_piv4), written one step ahead
    __shared__ __attribute__((aligned(16))) double s_E[2][16];   // [step parity] the next pivot block before the step's update
#ifdef BA_DIAG_STAMPS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
    if (tid == 0) { st->dbg[0] = tprev; st->dbg[1] = __builtin_amdgcn_s_memrealtime(); }
#endif
    {   // the band arrives in its LDS layout (k_schur_reduce wrote it that way): a straight copy, 16 bytes per lane and load
        typedef __attribute__((address_space(1))) double2_t gdouble2;
        const gdouble2* src = (const gdouble2*)W.Sb;
        double2_t* dst = reinterpret_cast<double2_t*>(Ab);
        const int cnt = ((n + 1) * RS) >> 1;   // RS is even
        for (int i0 = tid; i0 < cnt; i0 += 9 * LD_THREADS) {
            double2_t v[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) { const int i = i0 + u * LD_THREADS; v[u] = i < cnt ? src[i] : (double2_t){0, 0}; }
#pragma unroll
            for (int u = 0; u < 9; ++u) { const int i = i0 + u * LD_THREADS; if (i < cnt) dst[i] = v[u]; }
        }
    }
    for (int i = tid; i < ldlt_band_ylen(n); i += LD_THREADS) y[i] = i < n ? W.rhs[i] : 0.0;
    __syncthreads();
    if (is_piv) {   // the first pivot block and the preview of the second
        double e[10], G[16];
        int q = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) { const double v = Ab[min(i, n) * RS + j - i + bw + 3]; e[q++] = i < n ? v : 0.0; }
        bool bad;
        ldlt_piv4(e, min(4, n), G, bad);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) reinterpret_cast<double2_t*>(fac)[i] = (double2_t){G[2 * i], G[2 * i + 1]};
            if (bad) *s_fail = 1;
        }
        if (lane < 16) {
            const int i = lane >> 2, j = lane & 3, r = 4 + i;
            const double v = Ab[min(r, n) * RS + j - i + bw + 3];
            s_E[0][lane] = (j <= i && r < n) ? v : 0.0;
        }
    }
    __syncthreads();
#ifdef BA_DIAG_STAMPS
    { unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[0] += tn - tprev; tprev = tn; }
#endif
    const int nb16 = (n + 15) >> 4;
    const int li = lane & 15, lk = lane >> 4;
    const int ZA = n * RS;            // four zeros (the extra row): where a lane without an operand reads
    __shared__ double s_dump[64];
    const int dump = (int)(s_dump - sm) + lane;   // (a double index relative to sm, like every other offset here)
    const int BIG = 0x40000000;
    double4_t acc[3];
    // Tile (I, J) of tile diagonal d = I - J:  d <= 2 belongs to the ROW wave I mod 5 (cell d; the wave's tiles share one A
    // operand), d = 3, 4 -- the window's lower left corner, three tiles at a time -- to the HELPER wave (cell 0: d = 4, cells
    // 1, 2: d = 3 with J even / odd; an A operand per cell).  No update wave has more than three tiles at any step.
    // Per cell: the tile (scalars), the lane's operand addresses at k = 0 (they advance with k), "row - 4" / "column - 4" for
    // the reach test 0 <= x - 4 - k < bw (BIG: never), where the lane's four accumulator entries live in the band, which of
    // them exist (shape of the band x inside the matrix).  A cell whose tile has fallen behind the pivot adopts the next one
    // of its sequence -- constants on the addresses, the band's shape stays -- and reads it from the band, where nothing has
    // touched it yet: it comes into reach a step or more later (bw <= 59).
    int cIt[3], cJt[3], cra[3], crt[3], cba[3], cbt[3], cwa[3], cwm[3], csh[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { acc[c] = (double4_t){0, 0, 0, 0}; cIt[c] = 0; cJt[c] = -1; cra[c] = 0; crt[c] = BIG; cba[c] = 0; cbt[c] = BIG; cwa[c] = 0; cwm[c] = 0; csh[c] = 0; }
    // Every role runs its own copy of the step loop (same number of barriers): no role dispatch inside a step, and the
    // update code is compiled once for the row waves and once for the helper.
#ifdef BA_DIAG_STAMPS
#define LB_STEP_BEGIN const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
#define LB_STEP_END { __builtin_amdgcn_s_waitcnt(0); ph[3] += __builtin_amdgcn_s_memtime() - tb0; }   // busy part of the step, per wave
#else
#define LB_STEP_BEGIN
#define LB_STEP_END
#endif
    auto update_step = [&](int k, auto hlp_c) {
        constexpr bool HLP = decltype(hlp_c)::value;
        const int par = (k >> 2) & 1;
            const int Jlo = k >> 4;
            const int Ihi = min((k + 3 + bw) >> 4, nb16 - 1);
            const int Jn = (k + 4) >> 4, Jp = (k + 8) >> 4;
            const unsigned jn = (unsigned)(li - ((k + 4) & 15)), jp = (unsigned)(li - ((k + 8) & 15));
            const int op = (k + 8) & 15;
            if ((k & 15) == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    // row wave: all three cells follow the row (I += 5); helper: cell 0 moves by one block, cells 1, 2 by two
                    const int adv = HLP ? (c == 0 ? 1 : 2) : 5;
                    const bool gone = HLP ? __builtin_amdgcn_readfirstlane(cJt[c]) < Jlo : __builtin_amdgcn_readfirstlane(cIt[c]) < Jlo;
                    if (k != 0 && !gone) continue;   // (wave-uniform)
                    if (k == 0) {
                        const int It = HLP ? (c == 1 ? 3 : 4) : uw, Jt = HLP ? (c == 2 ? 1 : 0) : uw - c;
                        const int r0 = 16 * It + lk, cc = 16 * Jt + li;
                        cIt[c] = It; cJt[c] = Jt;
                        cra[c] = (16 * It + li) * (RS - 1) + bw + 3;
                        cba[c] = cc * (RS - 1) + lk + bw + 3;
                        cwa[c] = r0 * (RS - 1) + cc + bw + 3;
                        int m = 0;
#pragma unroll
                        for (int g = 0; g < 4; ++g) if ((unsigned)(r0 + 4 * g - cc) <= (unsigned)bw) m |= 1 << g;
                        csh[c] = m;
                    } else {
                        cIt[c] += adv; cJt[c] += adv;
                        cra[c] += 16 * adv * (RS - 1); cba[c] += 16 * adv * (RS - 1); cwa[c] += 16 * adv * RS;
                    }
                    const int r = 16 * cIt[c] + li, r0 = 16 * cIt[c] + lk, cc = 16 * cJt[c] + li;
                    crt[c] = r < n ? r - 4 : BIG;
                    const bool cok = cc >= 0 && cc < n;
                    cbt[c] = cok ? cc - 4 : BIG;
                    const int nrow = min(max((n - r0 + 3) >> 2, 0), 4);   // rows r0 + 4 g inside the matrix
                    const int m = cok ? csh[c] & ((1 << nrow) - 1) : 0;
                    cwm[c] = m;
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[c][g] = Ab[(m >> g & 1) ? cwa[c] + 4 * g * (RS - 1) : ZA];
                }
            }
            bool act[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) { const int It = __builtin_amdgcn_readfirstlane(cIt[c]), Jt = __builtin_amdgcn_readfirstlane(cJt[c]); act[c] = Jt >= Jlo && It <= Ihi; }
            if (act[0] || act[1] || act[2]) {
                const double2_t ga = reinterpret_cast<const double2_t*>(fac + 4 * k)[2 * lk], gb = reinterpret_cast<const double2_t*>(fac + 4 * k)[2 * lk + 1];
                double bvl[3], av[3];
                if (HLP) {
                    double q0[3], q1[3], q2[3], q3[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {   // (only the cells in the window: one of three when bw < 49)
                        q0[c] = 0; q1[c] = 0; q2[c] = 0; q3[c] = 0; bvl[c] = 0;
                        if (!act[c]) continue;
                        const int aa = (unsigned)(crt[c] - k) < (unsigned)bw ? cra[c] + k : ZA;
                        q0[c] = Ab[aa]; q1[c] = Ab[aa + 1]; q2[c] = Ab[aa + 2]; q3[c] = Ab[aa + 3];
                        bvl[c] = Ab[(unsigned)(cbt[c] - k) < (unsigned)bw ? cba[c] + k : ZA];
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) { asm volatile("" : "+v"(q0[c]), "+v"(q2[c])); av[c] = 0; if (act[c]) av[c] = (q0[c] * ga.x + q1[c] * ga.y) + (q2[c] * gb.x + q3[c] * gb.y); }   // R_r (-G)
                } else {
                    const int aa = (unsigned)(crt[0] - k) < (unsigned)bw ? cra[0] + k : ZA;
                    const double q0 = Ab[aa], q1 = Ab[aa + 1], q2 = Ab[aa + 2], q3 = Ab[aa + 3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) bvl[c] = Ab[(unsigned)(cbt[c] - k) < (unsigned)bw ? cba[c] + k : ZA];
                    av[0] = (q0 * ga.x + q1 * ga.y) + (q2 * gb.x + q3 * gb.y);
                    av[1] = av[0]; av[2] = av[0];
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) asm volatile("" : "+v"(bvl[c]), "+v"(av[c]));   // (keeps every load up here: sunk into its tile's branch it would be waited for alone)
#pragma unroll
                for (int c = 0; c < 3; ++c)   // the MFMAs back to back (independent accumulators); their consumers follow
                    if (act[c]) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[c], bvl[c], acc[c], 0, 0, 0);
                if (k + 4 < n) {   // the next raw panel: the lanes whose column is one of k + 4 .. k + 7, in the tiles of block column Jn
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        if (act[c] && __builtin_amdgcn_readfirstlane(cJt[c]) == Jn && jn < 4u) {
#pragma unroll
                            for (int g = 0; g < 4; ++g) sm[(cwm[c] >> g & 1) ? cwa[c] + 4 * g * (RS - 1) : dump] = acc[c][g];   // entries outside the band: into the dump slots
                        }
                }
                // the pivot block after the next, as it stands now (rows op .. op + 3 of the diagonal tile (Jp, Jp): a row wave's cell 0, register op / 4)
                if (!HLP && k + 8 < n && act[0] && __builtin_amdgcn_readfirstlane(cJt[0]) == Jp) {
                    const double v = op == 0 ? acc[0][0] : op == 4 ? acc[0][1] : op == 8 ? acc[0][2] : acc[0][3];
                    if (jp < 4u) s_E[par ^ 1][4 * lk + (int)jp] = v;
                }
            }
    };
    auto pivot_step = [&](int k) {
        const int par = (k >> 2) & 1;
            if (k + 4 < n) {
                const int l16 = lane & 15, i = l16 >> 2, j = l16 & 3;
                const int nv = min(4, n - (k + 4));
                const double* Pi = Ab + (i < nv ? (k + 4 + i) * RS + (bw - 1 - i) : ZA);   // (rows behind n: zeros)
                const double* Pj = Ab + (j < nv ? (k + 4 + j) * RS + (bw - 1 - j) : ZA);
                double pi[4], pj[4], gj[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) { pi[m] = Pi[m]; pj[m] = Pj[m]; gj[m] = fac[4 * k + 4 * j + m]; }
                const double eij = s_E[par][4 * max(i, j) + min(i, j)];
                const double qij = (pi[0] * gj[0] + pi[1] * gj[1]) + (pi[2] * gj[2] + pi[3] * gj[3]);   // (P (-G))(i, j)
                double en = (eij + dpp_quad_d<0x00>(qij) * pj[0] + dpp_quad_d<0x55>(qij) * pj[1]) + (dpp_quad_d<0xAA>(qij) * pj[2] + dpp_quad_d<0xFF>(qij) * pj[3]);
                if (max(i, j) >= nv) en = i == j ? 1.0 : 0.0;   // a ragged last block: identity padding
                double e[10], G[16];
                e[0] = readlane_d(en, 0);
                e[1] = readlane_d(en, 4); e[2] = readlane_d(en, 5);
                e[3] = readlane_d(en, 8); e[4] = readlane_d(en, 9); e[5] = readlane_d(en, 10);
                e[6] = readlane_d(en, 12); e[7] = readlane_d(en, 13); e[8] = readlane_d(en, 14); e[9] = readlane_d(en, 15);
                bool bad;
                ldlt_piv4(e, 4, G, bad);
                if (lane == 0) {
                    double2_t* fo = reinterpret_cast<double2_t*>(fac + 4 * (k + 4));
#pragma unroll
                    for (int u = 0; u < 8; ++u) fo[u] = (double2_t){G[2 * u], G[2 * u + 1]};
                    if (bad) *s_fail = 1;
                }
            }
    };
    auto rhs_step = [&](int k) {
            double gy[4];
            {
                const double y0 = y[k], y1 = y[k + 1], y2 = y[k + 2], y3 = y[k + 3];
#pragma unroll
                for (int j = 0; j < 4; ++j) gy[j] = (fac[4 * k + 4 * j] * y0 + fac[4 * k + 4 * j + 1] * y1) + (fac[4 * k + 4 * j + 2] * y2 + fac[4 * k + 4 * j + 3] * y3);
            }
            const int c = k + 4 + lane;
            const bool cv = lane < bw && c < n;
            const int ca = cv ? c * RS + (k - c + bw + 3) : ZA;   // entries left of the band are the row's zero slots
            const double r0 = Ab[ca], r1 = Ab[ca + 1], r2 = Ab[ca + 2], r3 = Ab[ca + 3];
            const double yc = y[cv ? c : 0];
            if (cv) y[c] = yc + (r0 * gy[0] + r1 * gy[1] + r2 * gy[2] + r3 * gy[3]);   // gy = -G y_k
    };
    if (uw >= 0) {
        for (int k = 0; k < n; k += 4) { LB_STEP_BEGIN update_step(k, std::false_type{}); LB_STEP_END __syncthreads(); }
    } else if (is_hlp) {
        for (int k = 0; k < n; k += 4) { LB_STEP_BEGIN update_step(k, std::true_type{}); LB_STEP_END __syncthreads(); }
    } else if (is_piv) {
        for (int k = 0; k < n; k += 4) { LB_STEP_BEGIN pivot_step(k); LB_STEP_END __syncthreads(); }
    } else if (is_rhs) {
        for (int k = 0; k < n; k += 4) { LB_STEP_BEGIN rhs_step(k); LB_STEP_END __syncthreads(); }
    } else {
        for (int k = 0; k < n; k += 4) __syncthreads();
    }
#undef LB_STEP_BEGIN
#undef LB_STEP_END
#ifdef BA_DIAG_STAMPS
    { unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[1] += tn - tprev; tprev = tn; }
#endif
    if (*s_fail) { if (tid == 0) st->ok2 = 0; return; }
    // Block back-substitution, right-looking, on ONE wave: with v = y (the block forward substitution), for the pivot blocks
    // s from the last to the first:  x_s = G_s v_s,  then  v_c -= sum_j R(4 s + j, c) x_{s, j}  for the columns c left of the
    // block inside the band -- rows 4 s .. 4 s + 3 of the raw band, one column per lane, no reduction across lanes.  Lane l
    // holds the column c = l (mod 64) of the 64 below the block in a register; a column takes its y the step it comes into
    // reach (distance d = 4 s - 1 - c <= bw - 1).  The block's own quad multiplies by G (quad broadcasts), v_readlane
    // hands x to every lane.
    if (wv == 0) {
        const int S4 = (n + 3) >> 2;
        const int lj = lane & 3;
        double v;
        { const int d = (4 * S4 - 1 - lane) & 63, c = 4 * S4 - 1 - d; v = y[max(c, 0)]; if (c < 0) v = 0.0; }
        const double2_t* gp = reinterpret_cast<const double2_t*>(fac + 4 * lj);
        // Everything a block needs from LDS is fetched one block ahead (y of the entering columns, the four band rows, G's row),
        // into two sets of registers used alternately.  The lane's column, its distance to the block and the address of its
        // entry in the block's first row move by constants from block to block: the column stays (d -= 4) until it becomes
        // part of the block itself, then the lane takes the column 64 below (d += 60).
        struct Ops { double ye, r0, r1, r2, r3; double2_t ga, gb; bool enter; };
        int fd, fc, fra;
        { const int k = 4 * (S4 - 1); fd = (k - 1 - lane) & 63; fc = k - 1 - fd; fra = k * (RS - 1) + fc + bw + 3; }
        auto fetch = [&](Ops& O, int sb, bool ragged) {
            const bool reach = fd <= bw - 1 && fc >= 0;
            O.enter = fd >= bw - 4 && reach;
            const int a0 = reach ? fra : ZA, st = reach ? RS - 1 : 0;   // R(k, c); rows k + 1 .. k + 3 follow at RS - 1 each
            O.ye = y[max(fc, 0)];
            O.r0 = Ab[a0];
            if (!ragged) { O.r1 = Ab[a0 + st]; O.r2 = Ab[a0 + 2 * st]; O.r3 = Ab[a0 + 3 * st]; }
            else {   // the last pivot block of a system whose size is no multiple of four: rows >= n read zeros
                const int k = 4 * sb;
                O.r1 = Ab[k + 1 < n ? a0 + st : ZA]; O.r2 = Ab[k + 2 < n ? a0 + 2 * st : ZA]; O.r3 = Ab[k + 3 < n ? a0 + 3 * st : ZA];
            }
            O.ga = gp[8 * sb]; O.gb = gp[8 * sb + 1];
        };
        auto advance = [&]() {
            const bool wrap = fd < 4;
            fd = wrap ? fd + 60 : fd - 4;
            fc = wrap ? fc - 64 : fc;
            fra -= 4 * (RS - 1) + (wrap ? 64 : 0);
        };
        auto step = [&](const Ops& O, int sb) {
            const int k = 4 * sb;
            const double v0 = dpp_quad_d<0x00>(v), v1 = dpp_quad_d<0x55>(v), v2 = dpp_quad_d<0xAA>(v), v3 = dpp_quad_d<0xFF>(v);
            const double x = -((O.ga.x * v0 + O.ga.y * v1) + (O.gb.x * v2 + O.gb.y * v3));
            const int q0 = 4 * (sb & 15);
            const double x0 = readlane_dyn_d(x, q0), x1 = readlane_dyn_d(x, q0 + 1), x2 = readlane_dyn_d(x, q0 + 2), x3 = readlane_dyn_d(x, q0 + 3);
            if ((lane & 60) == q0 && k + lj < n) W.rhs[k + lj] = x;   // the solution leaves from here
            if (O.enter) v = O.ye;
            v -= (O.r0 * x0 + O.r1 * x1) + (O.r2 * x2 + O.r3 * x3);
        };
        Ops A, B;
        int sb = S4 - 1;
        fetch(A, sb, true);
        while (true) {
            if (sb > 0) { advance(); fetch(B, sb - 1, false); }
            step(A, sb);
            if (--sb < 0) break;
            if (sb > 0) { advance(); fetch(A, sb - 1, false); }
            step(B, sb);
            if (--sb < 0) break;
        }
    }
    __syncthreads();
#ifdef BA_DIAG_STAMPS
    { unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[5] += tn - tprev; tprev = tn; }
    if (lane == 0 && is_piv) st->dbg[7] = ph[3];
    if (lane == 0 && is_rhs) st->dbg[6] = ph[3] << 32;   // (tid 0 ors its part in below: after the barrier of the last loop)
    __syncthreads();
    if (tid == 0) { st->dbg[2] = __builtin_amdgcn_s_memtime(); st->dbg[3] = __builtin_amdgcn_s_memrealtime();
                    st->dbg[4] = (ph[0] << 32) | ph[1]; st->dbg[5] = (ph[2] << 32) | ph[3]; st->dbg[6] |= ph[5]; }
#endif
#ifdef BA_DIAG_WAVES
    __syncthreads();
    if (lane == 0) st->dbg[wv] = ph[3];   // every wave's busy cycles in the factor loop
#endif
    if (tid == 0) st->ok2 = 1;
}

// The reduced solve of a window is ONE of two kernels, chosen on the host from the window's size and envelope
// (BaWin::solver, ba_api.hip); a batch launches the kinds it contains and a workgroup whose window is of the other kind
// returns at once.  (As one kernel with a run-time branch the register allocation was the union of the paths: 256 VGPRs
// and 37 spilled; on its own the banded kernel takes 94.)  Each starts with k_iter_begin's bookkeeping for the slots without that launch (bak_slot, first ==
// false): it is the first single-workgroup kernel behind the last reader of need_linearize (k_pose_reduce).
__global__ __launch_bounds__(LD_THREADS) void k_ldlt_band(BaWin* wins) {
    const BaWin& W = wins[blockIdx.y];
    BaState* st = BA_ST(wins, blockIdx.y);
    if (W.solver != BA_SOLVER_BAND || st->done) return;
    if (threadIdx.x == 0 && st->need_linearize) { st->iniChi = st->currentChi; st->qmax = 0; st->need_linearize = 0; }
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ int s_fail;
    if (threadIdx.x == 0) s_fail = 0;
    __syncthreads();
    ldlt_band_solve(W, st, sm, &s_fail);
}

__global__ __launch_bounds__(LD_THREADS) void k_ldlt_blocked(BaWin* wins) {
    const BaWin& W = wins[blockIdx.y];
    BaState* st = BA_ST(wins, blockIdx.y);
    if (W.solver != BA_SOLVER_BLOCKED || st->done) return;
    const int n = W.nS, N = W.Npad;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0 && st->need_linearize) { st->iniChi = st->currentChi; st->qmax = 0; st->need_linearize = 0; }
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* Dg = sm;                      // LD_NB x LD_P: diagonal block (unit L below, D on the diagonal)
    double* Wd = Dg + LD_NB * LD_P;       // (rows below + rhs row, padded to 16) x LD_P: L21 * D
    __shared__ int s_fail;
    __shared__ double s_invd[LD_NB], s_corr[LD_NB], s_dval[LD_NB];
    if (tid == 0) s_fail = 0;
    if (n == 0) { if (tid == 0) st->ok2 = 1; return; }
#ifdef BA_DIAG_STAMPS
    unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
#define STAMP(i) do { unsigned long long tn = __builtin_amdgcn_s_memtime(); ph[i] += tn - tprev; tprev = tn; } while (0)
    if (tid == 0) { st->dbg[0] = tprev; st->dbg[1] = __builtin_amdgcn_s_memrealtime(); }
#else
#define STAMP(i)
#endif
    gdouble* S = (gdouble*)W.S;
    for (int i = tid; i < n; i += LD_THREADS) S[(size_t)n * N + i] = W.rhs[i];  // rhs as row n
    __syncthreads();
    for (int jb = 0; jb < n; jb += LD_NB) {
        const int nb = min(LD_NB, n - jb);
        const int base = jb + nb;
        // rows under the panel that can hold an entry in its columns (row envelope, ba_api.hip); LDLt without pivoting
        // never fills outside the envelope, so the rows beyond keep exact zeros there and are not touched
        const int below = max((int)W.panel_hi[jb / LD_NB] + 1 - base, 0);
        const int rows = below + 1;          // + the rhs row, which is matrix row n
#define GROW(r) ((r) < below ? base + (r) : n)
        const int rows16 = (rows + 15) & ~15;
        // The staging and write-back loops of a panel index with `tp`, the thread id behind an empty asm: their dozens of LDS / matrix
        // addresses are then formed per panel (an add and a shift each) instead of being hoisted out of the panel loop, where they
        // had to live in registers across the whole kernel -- and were spilled to scratch and reloaded per panel (27 VGPRs).
        int tp = tid;
        asm volatile("" : "+v"(tp));
        for (int i = tp; i < nb * LD_NB; i += LD_THREADS) {
            const int r = i >> 5, c = i & 31;
            if (c < nb) Dg[r * LD_P + c] = S[(size_t)(jb + r) * N + jb + c];
        }
        if (nb < LD_NB)
            for (int i = tp; i < LD_NB * LD_NB; i += LD_THREADS) {
                const int r = i >> 5, c = i & 31;
                if (r >= nb || c >= nb) Dg[r * LD_P + c] = 0.0;
            }
        for (int i0 = tp; i0 < rows16 * LD_NB; i0 += 8 * LD_THREADS) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {   // issue all loads first: the L2 round trip is paid once per batch
                const int i = i0 + u * LD_THREADS, r = i >> 5, c = i & 31;
                v[u] = (i < rows16 * LD_NB && r < rows && c < nb) ? S[(size_t)GROW(r) * N + jb + c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * LD_THREADS, r = i >> 5, c = i & 31;
                if (i < rows16 * LD_NB) Wd[r * LD_P + c] = v[u];
            }
        }
        __syncthreads();
        STAMP(0);
        ldlt_factor_diag(Dg, s_invd, s_corr, s_dval, nb, tid, &s_fail);
        STAMP(1);
        __syncthreads();
        if (s_fail) break;
        ldlt_rows(Wd, Dg, rows, tid);
        __syncthreads();
        STAMP(2);
        // write back the factored panel: L11 / D, and L21 = (L*d) / d
        for (int i = tp; i < nb * LD_NB; i += LD_THREADS) {
            const int r = i >> 5, c = i & 31;
            if (c <= r) S[(size_t)(jb + r) * N + jb + c] = Dg[r * LD_P + c];
        }
        for (int i = tp; i < rows * LD_NB; i += LD_THREADS) {
            const int r = i >> 5, c = i & 31;
            if (c < nb) S[(size_t)GROW(r) * N + jb + c] = Wd[r * LD_P + c] * s_invd[c];
        }
        // trailing update on MFMA: C[r][c] -= sum_k (w[r][k] invd[k]) w[c][k],  c <= r
        STAMP(3);
#ifndef LD_DIAG_SKIP_TRAIL
        const int RT = rows16 >> 4, CT = (below + 15) >> 4;
        // lower-triangular tile list in row-major order: t -> (rt, ct), ct <= min(rt, CT-1).  Each wave takes a
        // contiguous run of it, so the A fragments (-w[r][k] / d_k, 8 doubles per lane) are loaded once per row tile
        // and a tile costs 8 LDS reads (B fragments) + 8 MFMAs; the next tile's C (L2) and B (LDS) are fetched while
        // the current tile's MFMAs run.
        int ntile = 0;
        for (int rt = 0; rt < RT; ++rt) ntile += min(rt + 1, CT);
        const int per = (ntile + LD_THREADS / 64 - 1) / (LD_THREADS / 64);
        int t = wv * per;
        const int tend = min(ntile, t + per);
        if (t < tend) {
            int rt = 0, ct = t;
            while (ct >= min(rt + 1, CT)) { ct -= min(rt + 1, CT); ++rt; }
            const int l15 = lane & 15, lk = lane >> 4;
            double invd8[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) invd8[ks] = s_invd[4 * ks + lk];
            // LD_CDEPTH C tiles are in flight from L2 per wave; S is addressed in the global address space, so these
            // loads sit in vmcnt only and the B fragments' LDS waits pass them.  (Measured on the first panel, 20 tiles per
            // wave: ~2,400 cycles per tile = ~1,000 of MFMA (two waves share a SIMD's matrix unit), ~650 for the C loads
            // and ~400 for the stores -- one CU moves 64 B/clk and this panel's C traffic alone is 612 KB -- rest LDS.)
            double af[8], bf[8];
            double4_t cq[LD_CDEPTH];
            int art = -1;
            auto load_c = [&](int frt, int fct, double4_t& C) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int r = 16 * frt + lk + 4 * g, c = 16 * fct + l15;
                    C[g] = (r < rows && c < below && c <= r) ? S[(size_t)GROW(r) * N + base + c] : 0.0;
                }
            };
            auto advance = [&](int& art_, int& act_) { if (++act_ >= min(art_ + 1, CT)) { ++art_; act_ = 0; } };
            int lrt = rt, lct = ct;   // tile whose C is fetched next
#pragma unroll
            for (int d = 0; d < LD_CDEPTH; ++d)
                if (t + d < tend) { load_c(lrt, lct, cq[d]); advance(lrt, lct); }
            for (; t < tend; t += LD_CDEPTH) {
#pragma unroll
                for (int d = 0; d < LD_CDEPTH; ++d) {
                    if (t + d < tend) {
                        if (rt != art) {
#pragma unroll
                            for (int ks = 0; ks < 8; ++ks) af[ks] = -Wd[(16 * rt + l15) * LD_P + 4 * ks + lk] * invd8[ks];
                            art = rt;
                        }
#pragma unroll
                        for (int ks = 0; ks < 8; ++ks) bf[ks] = Wd[(16 * ct + l15) * LD_P + 4 * ks + lk];
                        double4_t acc = cq[d];
#pragma unroll
                        for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks], bf[ks], acc, 0, 0, 0);
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int r = 16 * rt + lk + 4 * g, c = 16 * ct + l15;
                            if (r < rows && c < below && c <= r) S[(size_t)GROW(r) * N + base + c] = acc[g];
                        }
                        if (t + d + LD_CDEPTH < tend) { load_c(lrt, lct, cq[d]); advance(lrt, lct); }
                        advance(rt, ct);
                    }
                }
            }
        }
#endif
        __syncthreads();
        STAMP(4);
#undef GROW
    }
    if (s_fail) {
        if (tid == 0) st->ok2 = 0;
        return;
    }
#ifndef LD_DIAG_SKIP_BACK
    // ---- backward substitution x = L^-T y, y = row n of S ----
    // Right-looking over 32-row blocks from the bottom: wavefront 0 solves the block's triangular
    // system (lane k owns x_k, columns of L11 in registers, v_readlane broadcasts), then every
    // earlier unknown i < jb takes y_i -= sum_m L[jb+m][i] * x_m, reading ROWS of L (coalesced).
    double* xs = Wd;                 // n doubles
    for (int i = tid; i < n; i += LD_THREADS) xs[i] = S[(size_t)n * N + i];
    // L is final now, so nothing a block needs depends on x: wavefront 0 fetches the NEXT block's L11 columns, and
    // every thread the next block's L rows, while the current block is being solved / applied.
    const int jb_last = ((n - 1) / LD_NB) * LD_NB;
    double col[LD_NB], lv[LD_NB];
    auto fetch_col = [&](int jb) {
        const int nb = min(LD_NB, n - jb), k = lane;
#pragma unroll
        for (int m = 0; m < LD_NB; ++m) col[m] = (k < nb && m < nb && m > k) ? S[(size_t)(jb + m) * N + jb + k] : 0.0;
    };
    auto fetch_rows = [&](int jb) {   // rows jb .. jb+nb-1 of L at column tid (columns >= LD_THREADS: plain loop below)
        const int nb = min(LD_NB, n - jb);
#pragma unroll
        for (int m = 0; m < LD_NB; ++m) lv[m] = (m < nb && tid < jb) ? S[(size_t)(jb + m) * N + tid] : 0.0;
    };
    if (wv == 0) fetch_col(jb_last);
    fetch_rows(jb_last);
    __syncthreads();
    for (int jb = jb_last; jb >= 0; jb -= LD_NB) {
        const int nb = min(LD_NB, n - jb);
        if (wv == 0) {
            const int k = lane;
            double v = (k < nb) ? xs[jb + k] : 0.0;
#pragma unroll
            for (int m = LD_NB - 1; m >= 0; --m) {
                const double xm = readlane_d(v, m);   // final once every higher index has been applied
                if (m < nb && k < m) v -= col[m] * xm;
            }
            if (k < nb) xs[jb + k] = v;
            if (jb >= LD_NB) fetch_col(jb - LD_NB);
        }
        __syncthreads();
        if (tid < jb) {
            double acc = xs[tid];
#pragma unroll
            for (int m = 0; m < LD_NB; ++m) acc -= lv[m] * xs[jb + m];
            xs[tid] = acc;
        }
        for (int i = tid + LD_THREADS; i < jb; i += LD_THREADS) {   // systems wider than the workgroup
            double acc = xs[i];
            for (int m = 0; m < nb; ++m) acc -= S[(size_t)(jb + m) * N + i] * xs[jb + m];
            xs[i] = acc;
        }
        if (jb >= LD_NB) fetch_rows(jb - LD_NB);
        __syncthreads();
    }
    for (int i = tid; i < n; i += LD_THREADS) W.rhs[i] = xs[i];
#endif
#ifdef BA_DIAG_STAMPS
    STAMP(5);
    if (tid == 0) { st->dbg[2] = __builtin_amdgcn_s_memtime(); st->dbg[3] = __builtin_amdgcn_s_memrealtime();
                    st->dbg[4] = (ph[0] << 32) | ph[1]; st->dbg[5] = (ph[2] << 32) | ph[3]; st->dbg[6] = (ph[4] << 32) | ph[5]; }
#endif
    if (tid == 0) st->ok2 = 1;
}

// ---- S9: landmark back-substitution, push(), oplus ------------------------------------------------
__global__ __launch_bounds__(256) void k_backsub_update(BaWin* wins) {   // BA_PG lanes per point; thread t < n_kf also moves pose t
    const BaWin& W = wins[blockIdx.y];
    BaState* st = BA_ST(wins, blockIdx.y);
    if (st->done) return;
    __shared__ double sh[4];
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int p = t / BA_PG, g = t % BA_PG;
    const bool ok = st->ok2 != 0;
    const double lambda = st->lambda;
    const bool live = p < W.n_pt;
    double cl[3] = {0, 0, 0};
    if (live && ok) {
        for (int i = W.pt_ptr[p] + g; i < W.pt_ptr[p + 1]; i += BA_PG) {
            const int e = W.pt_edges[i];
            if (!W.e_active[e]) continue;
            const int col = W.pose_col[W.e_kf[e]];
            if (col < 0) continue;
            double H[18];
            if (W.nrow == 3) hpl_of3(W.e_jac + BA_JAC_STEREO * (size_t)e, H);
            else hpl_of(W.e_jac + 21 * (size_t)e, H);
            const double* xp = W.rhs + 6 * col;
            for (int j = 0; j < 3; ++j)
                for (int r = 0; r < 6; ++r) cl[j] -= H[3 * r + j] * xp[r];
        }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) cl[j] = group_sum(cl[j]);
    double scale = 0;
    if (live && g == 0) {
        const double* bl = W.bl + 3 * (size_t)p;
        double xl[3] = {0, 0, 0};
        if (ok) {
            for (int j = 0; j < 3; ++j) cl[j] += bl[j];
            const double* Di = W.Dinv + 6 * (size_t)p;
            xl[0] = Di[0] * cl[0] + Di[1] * cl[1] + Di[2] * cl[2];
            xl[1] = Di[1] * cl[0] + Di[3] * cl[1] + Di[4] * cl[2];
            xl[2] = Di[2] * cl[0] + Di[4] * cl[1] + Di[5] * cl[2];
        }
        for (int j = 0; j < 3; ++j) {
            W.x_l[3 * (size_t)p + j] = xl[j];
            scale += xl[j] * (lambda * xl[j] + bl[j]);
            const double v = W.pt[3 * (size_t)p + j];
            W.pt_bak[3 * (size_t)p + j] = v;
            W.pt[3 * (size_t)p + j] = v + xl[j];
        }
    }
    if (t < W.n_kf) {
        double T[7];
        for (int i = 0; i < 7; ++i) { T[i] = W.pose[7 * (size_t)t + i]; W.pose_bak[7 * (size_t)t + i] = T[i]; }
        const int col = W.pose_col[t];
        if (col >= 0 && ok) {
            pose_oplus(T, W.rhs + 6 * col);
            for (int i = 0; i < 7; ++i) W.pose[7 * (size_t)t + i] = T[i];
        }
    }
    const double tot = block_sum_256(scale, sh);
    if (threadIdx.x == 0) W.scale_part[blockIdx.x] = tot;
}

// ---- S10: residuals at the tentative state, robust cost partial sums ------------------------------------
__global__ __launch_bounds__(256) void k_errors(BaWin* wins) {
    const BaWin& W = wins[blockIdx.y];
    BaState* st = BA_ST(wins, blockIdx.y);
    if (st->done) return;
    __shared__ double sh[4];
    const int e = blockIdx.x * 256 + threadIdx.x;
    double rho = 0;
    if (e < W.n_edge && W.e_active[e]) {
        if (W.nrow == 3) {
            const EdgeGeom3 g = edge_eval3(W, e);
            W.e_chi2[e] = g.chi2;
            const double delta = g.stereo ? W.huber_delta_s : W.huber_delta, dsqr = huber_dsqr(delta);
            if (st->robust && g.chi2 > dsqr) rho = 2 * sqrt(g.chi2) * delta - dsqr;
            else rho = g.chi2;
        } else {
        const EdgeGeom g = edge_eval(W, e);
        W.e_chi2[e] = g.chi2;
        const double dsqr = huber_dsqr(W.huber_delta);
        if (st->robust && g.chi2 > dsqr) rho = 2 * sqrt(g.chi2) * W.huber_delta - dsqr;
        else rho = g.chi2;
        }
    }
    const double tot = block_sum_256(rho, sh);
    if (threadIdx.x == 0) W.chi_part[blockIdx.x] = tot;
}

__device__ double sum_parts(const double* part, int n, double* sh) {
    double v = 0;
    for (int i = threadIdx.x; i < n; i += 256) v += part[i];
    return block_sum_256(v, sh);
}

// ---- S11: accept / reject, lambda update, stop rules (one workgroup per window) ----------------------------
__global__ __launch_bounds__(256) void k_decide(BaWin* wins) {
    const BaWin& W = wins[blockIdx.y];
    BaState* st = BA_ST(wins, blockIdx.y);
    if (st->done) return;
    __shared__ double sh[4];
    __shared__ int s_reject;
    const int tid = threadIdx.x;
    const int ne_blocks = (W.n_edge + 255) / 256, np_blocks = (W.n_pt * BA_PG + 255) / 256;
    double tempChi = sum_parts(W.chi_part, ne_blocks, sh);
    double scale = sum_parts(W.scale_part, np_blocks, sh);
    double ps = 0;
    for (int i = tid; i < W.nS; i += 256) { const double x = st->ok2 ? W.rhs[i] : 0.0; ps += x * (st->lambda * x + W.bp[i]); }
    scale += block_sum_256(ps, sh);
    if (tid == 0) {
        if (!st->ok2) tempChi = DBL_MAX;
        st->tempChi = tempChi;
        double rho = (st->currentChi - tempChi) / (scale + 1e-3);
        const bool good = rho > 0 && fabs(tempChi) <= DBL_MAX;
        if (good) {
            double alpha = 1. - pow((2 * rho - 1), 3);
            alpha = fmin(alpha, 2. / 3.);
            const double sf = fmax(1. / 3., alpha);
            st->lambda *= sf;
            st->ni = 2;
            st->currentChi = tempChi;
        } else {
            st->lambda *= st->ni;
            st->ni *= 2;
        }
        s_reject = !good;
        st->qmax += 1;
        const bool again = rho < 0 && st->qmax < 10;
        if (!again) {  // the iteration is over
            const int s = st->stage, it = st->it;
            if (it < BA_MAX_ITS) { st->chi2[s][it] = tempChi; st->lam[s][it] = st->lambda; st->trials[s][it] = st->qmax; }
            st->n_its[s] = it + 1;
            bool terminate = st->qmax == 10 || rho == 0;
            if (!terminate) {
                if ((st->iniChi - st->currentChi) * 1e3 < st->iniChi) st->nBad += 1; else st->nBad = 0;
                if (st->nBad >= 3) terminate = true;
            }
            st->it = it + 1;
            st->need_linearize = 1;
            if (terminate || st->it >= st->max_it) st->done = 1;
        }
    }
    __syncthreads();
    if (s_reject) {  // pop(): restore the vertices
        for (int i = tid; i < 3 * W.n_pt; i += 256) W.pt[i] = W.pt_bak[i];
        for (int i = tid; i < 7 * W.n_kf; i += 256) W.pose[i] = W.pose_bak[i];
    }
}

// ---- stage control ---------------------------------------------------------------------------------------
// gate between the stages (Optimizer.cc:672-686): chi2 > gate || depth <= 0 -> level 1; kernels off
__global__ __launch_bounds__(256) void k_gate(BaWin* wins) {
    const BaWin& W = wins[blockIdx.y];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= W.n_edge) return;
    const EdgeGeom g = edge_eval(W, e);
    const bool out = W.e_chi2[e] > (edge_is_stereo(W, e) ? W.chi2_gate_s : W.chi2_gate) || !(g.z > 0.0);   // Optimizer.cc:680, :696
    W.e_out1[e] = out;
    if (out && W.e_active[e]) {
        W.e_active[e] = 0;
        // the edge leaves the Schur operands: k_prepare only (re)writes the blocks of active edges
        const int col = W.pose_col[W.e_kf[e]];
        if (col >= 0) {
            const size_t K = (size_t)W.Kpad;
            for (int r = 0; r < 6; ++r) {
                const size_t o = (size_t)(6 * col + r) * K + 3 * (size_t)W.e_pt[e];
                W.GA[o] = 0.0; W.GA[o + 1] = 0.0; W.GA[o + 2] = 0.0;
            }
        }
    }
}

// Zeroes the k ranges of the Schur operand the window's structure can touch (a solve starts from whatever the slab held before):
// per row tile the union of its range as the product's A operand and as its B operand (the tile with the right-hand side's row:
// every point).  Grid (x, row tile, window), one double2 per thread and step.
__global__ __launch_bounds__(256) void k_zero_operands(BaWin* wins) {
    const BaWin& W = wins[blockIdx.z];
    const int t = blockIdx.y;
    if (t >= W.Npad / BA_TILE) return;
    const size_t K = (size_t)W.Kpad;
    const bool ha = W.tile_ahi[t] > W.tile_alo[t], hb = W.tile_bhi[t] > W.tile_blo[t];
    if (!ha && !hb) return;
    const int lo = ha && hb ? min(W.tile_alo[t], W.tile_blo[t]) : ha ? W.tile_alo[t] : W.tile_blo[t];
    const int hi = ha && hb ? max(W.tile_ahi[t], W.tile_bhi[t]) : ha ? W.tile_ahi[t] : W.tile_bhi[t];
    gdouble* M = (gdouble*)W.GA + (size_t)(BA_TILE * t) * K;
    const int wdt = (hi - lo) >> 1;   // double2 per row (ranges are multiples of BA_KC)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)BA_TILE * wdt; i += (long)gridDim.x * 256) {
        const int r = (int)(i / wdt), c = (int)(i - (long)r * wdt);
        gdouble* q = M + (size_t)r * K + lo + 2 * c;
        q[0] = 0.0; q[1] = 0.0;
    }
}

// stage entry: counts the active edges; the following k_errors + k_stage_begin2 set the start cost
__global__ __launch_bounds__(256) void k_stage_begin(BaWin* wins, int stage, int max_it, int robust) {
    const BaWin& W = wins[blockIdx.y];
    BaState* st = BA_ST(wins, blockIdx.y);
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int c = 0;   // flags are 0 / 1 and the array starts on a 256-byte boundary (carve_work): 16 per load
    const uint4* a16 = reinterpret_cast<const uint4*>(W.e_active);
    const int full = W.n_edge >> 4;
    for (int i = threadIdx.x; i < full; i += 256) {
        const uint4 v = a16[i];
        c += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
    }
    for (int e = 16 * full + threadIdx.x; e < W.n_edge; e += 256) c += W.e_active[e] != 0;
    if (c) atomicAdd(&cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) {
        st->stage = stage; st->max_it = max_it; st->robust = robust;
        st->it = 0; st->qmax = 0; st->nBad = 0; st->need_linearize = 1; st->ok2 = 1;
        st->n_active = cnt;
        st->n_its[stage] = 0;
        st->lambda = -1; st->ni = 2;
        st->done = (cnt == 0 || max_it <= 0) ? 1 : 0;  // "0 vertices to optimize" / no iterations
    }
}
__global__ __launch_bounds__(256) void k_stage_begin2(BaWin* wins) {
    const BaWin& W = wins[blockIdx.y];
    BaState* st = BA_ST(wins, blockIdx.y);
    if (st->done) return;
    __shared__ double sh[4];
    const double chi = sum_parts(W.chi_part, (W.n_edge + 255) / 256, sh);
    if (threadIdx.x == 0) { st->currentChi = chi; st->chi2_init[st->stage] = chi; }
}

// final erasure test on every edge (Optimizer.cc:715-728), pose export as R|t, and every other output of the window
// copied into its contiguous output section (one device-to-host copy per window)
__global__ __launch_bounds__(256) void k_final(BaWin* wins, const BaIo* io) {
    const BaWin& W = wins[blockIdx.y];
    const BaIo& O = io[blockIdx.y];
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < W.n_edge) {
        const EdgeGeom g = edge_eval(W, e);
        O.out_flag[e] = W.e_chi2[e] > (edge_is_stereo(W, e) ? W.chi2_gate_s : W.chi2_gate) || !(g.z > 0.0);   // Optimizer.cc:723, :740
        O.out_chi2[e] = W.e_chi2[e];
        O.out_out1[e] = W.e_out1[e];
    }
    if (e < 3 * W.n_pt) O.out_pt[e] = W.pt[e];
    if (e < W.n_kf) {
        double R[9];
        quat_to_R(W.pose + 7 * (size_t)e, R);
        double* o = O.out_pose + 12 * (size_t)e;
        for (int i = 0; i < 9; ++i) o[i] = R[i];
        for (int i = 0; i < 3; ++i) o[9 + i] = W.pose[7 * (size_t)e + 4 + i];
    }
    if (e < (int)(sizeof(BaState) / 8)) reinterpret_cast<unsigned long long*>(O.out_state)[e] = reinterpret_cast<const unsigned long long*>(W.st)[e];
}

// window set-up on the device: input poses R|t -> normalised quaternion (Converter::toSE3Quat -> SE3Quat(R,t)), points into
// their working array, every edge active, LM state cleared
__global__ __launch_bounds__(256) void k_import(BaWin* wins, const BaIo* io) {
    const BaWin& W = wins[blockIdx.y];
    const BaIo& I = io[blockIdx.y];
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < W.n_edge) { W.e_active[k] = 1; W.e_out1[k] = 0; W.e_chi2[k] = 0.0; }
    if (k < 3 * W.n_pt) W.pt[k] = I.in_pt[k];
    if (k < (int)(sizeof(BaState) / 8)) reinterpret_cast<unsigned long long*>(W.st)[k] = 0ull;
    if (W.solver == BA_SOLVER_BAND)   // the zeros of the band image (pad slots, the part of a row left of column 0, the extra row)
        for (int i = k; i < (W.nS + 1) * ldlt_band_rs(W.band); i += gridDim.x * 256) W.Sb[i] = 0.0;
    if (k >= W.n_kf) return;
    const double* p = I.in_pose + 12 * (size_t)k;
    double R[9], q[4];
    for (int i = 0; i < 9; ++i) R[i] = p[i];
    R_to_quat(R, q);
    quat_normalize(q);
    double* T = W.pose + 7 * (size_t)k;
    T[0] = q[0]; T[1] = q[1]; T[2] = q[2]; T[3] = q[3]; T[4] = p[9]; T[5] = p[10]; T[6] = p[11];
}

// ---- launch wrappers -----------------------------------------------------------------------------------------
size_t bak_ldlt_smem(int Npad) {   // the blocked path's need, raised to the banded path's budget (one 512-thread workgroup per CU either way)
    return std::max(sizeof(double) * ((size_t)LD_NB * LD_P + ((size_t)Npad + 16) * LD_P), (size_t)LD_BAND_LDS);
}

hipError_t bak_prepare(int Npad) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ldlt_band), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bak_ldlt_smem(Npad));
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_ldlt_blocked), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bak_ldlt_smem(Npad));
    return e;
}

// the reduced solve a window takes (host): its structure decides
int bak_solver_kind(int n, int band) {
    static const bool no_band = getenv("SLAMIT_BA_NO_BAND") != nullptr;   // A/B and parity runs: no window through the banded kernel
    if (!no_band && ldlt_band_ok(n, band)) return BA_SOLVER_BAND;
    return BA_SOLVER_BLOCKED;
}

void bak_import(hipStream_t st, BaWin* wins, const BaIo* io, int max_kf, int max_pt, int max_edge, int Npad, int nwin) {
    const int nb = std::max(std::max(max_edge, 3 * max_pt), std::max(max_kf, (int)(sizeof(BaState) / 8)));
    hipLaunchKernelGGL(k_import, dim3((nb + 255) / 256, nwin), dim3(256), 0, st, wins, io);
    hipLaunchKernelGGL(k_zero_operands, dim3(64, Npad / BA_TILE, nwin), dim3(256), 0, st, wins);
}

void bak_stage_begin(hipStream_t st, BaWin* wins, int nwin, int max_edge, int stage, int max_it, int robust, bool gate) {
    const dim3 ge((max_edge + 255) / 256, nwin);
    if (gate) hipLaunchKernelGGL(k_gate, ge, dim3(256), 0, st, wins);
    hipLaunchKernelGGL(k_stage_begin, dim3(1, nwin), dim3(256), 0, st, wins, stage, max_it, robust);
    hipLaunchKernelGGL(k_errors, ge, dim3(256), 0, st, wins);
    hipLaunchKernelGGL(k_stage_begin2, dim3(1, nwin), dim3(256), 0, st, wins);
}

// split-K of the tiled Schur product (gridDim.y of its launch): a batch brings its own parallelism (64 windows: 2 / 4 / 8 / 16 splits -> 47.8k /
// 49.7k / 52.8k / 50.0k LM it/s).  (SLAMIT_BA_NSPLIT: A/B runs.)
int bak_nsplit(int nwin) {
    static const int nsplit_env = getenv("SLAMIT_BA_NSPLIT") ? atoi(getenv("SLAMIT_BA_NSPLIT")) : 0;
    return nsplit_env >= 1 && nsplit_env <= BA_SPLITS ? nsplit_env : nwin >= 16 ? 8 : BA_SPLITS;
}

// one LM trial slot for every window of the batch
// (`first`: the first slot of a stage, whose lambda comes out of the reductions; every later slot runs them and the
// damping in one launch, the pose blocks ride with the Schur product, and the iteration bookkeeping is left to the reduced solve's kernel: 7 launches
// instead of 11)
// `ev` (profiling solves only, slamit_ba_profile): six events recorded at the phase boundaries of the slot -- before the
// linearisation, after it, after the Schur complement, after the reduced solve, after the update, after residuals + decision
void bak_slot(hipStream_t st, BaWin* wins, int nwin, int max_kf, int max_pt, int max_edge, int Npad, bool first, unsigned solvers, hipEvent_t* ev) {
    const int nsplit = bak_nsplit(nwin);
    const dim3 ge((max_edge + 255) / 256, nwin), gp((max_pt * BA_PG + 255) / 256, nwin);
    if (ev) (void)hipEventRecord(ev[0], st);
    if (first) {
        hipLaunchKernelGGL(k_linearize, ge, dim3(256), 0, st, wins);
        hipLaunchKernelGGL(k_point_reduce, gp, dim3(256), 0, st, wins);
        hipLaunchKernelGGL(k_pose_reduce, dim3(max_kf, nwin), dim3(256), 0, st, wins);
        hipLaunchKernelGGL(k_iter_begin, dim3(1, nwin), dim3(64), 0, st, wins);
        hipLaunchKernelGGL(k_prepare, gp, dim3(256), 0, st, wins);
    } else {
        hipLaunchKernelGGL(k_point_pass, gp, dim3(256), 0, st, wins);
    }
    if (ev) (void)hipEventRecord(ev[1], st);
    const int T = Npad / BA_TILE, ntiles = T * (T + 1) / 2;
    // The pose blocks ride in the Schur launch of a single window (one launch less on its latency chain); a batch launches them on their own:
    // the fused kernel's register count is the pose reduction's (142 + 32: two waves per SIMD), the product alone runs three
    // (batch of 64, Schur phase: 2.97 -> 2.73 ms per 16 slots).
    if (!first && nwin >= 16) {
        hipLaunchKernelGGL(k_pose_reduce, dim3(max_kf, nwin), dim3(256), 0, st, wins);
        hipLaunchKernelGGL(k_schur, dim3(ntiles, nsplit, nwin), dim3(256), 0, st, wins);
    } else
    if (first) hipLaunchKernelGGL(k_schur, dim3(ntiles, nsplit, nwin), dim3(256), 0, st, wins);
    else hipLaunchKernelGGL(k_schur_pose, dim3(ntiles + (max_kf + nsplit - 1) / nsplit, nsplit, nwin), dim3(256), 0, st, wins, ntiles);
    // (a batch without a window of the blocked solver only needs the banded mapping's workgroups: four rows each)
    hipLaunchKernelGGL(k_schur_reduce, dim3((solvers & (1u << BA_SOLVER_BLOCKED)) ? (Npad * Npad + 255) / 256 : (Npad + 3) / 4, nwin), dim3(256), 0, st, wins, nsplit);
    if (ev) (void)hipEventRecord(ev[2], st);
    // `solvers`: bit BA_SOLVER_* set when a window of the batch takes that kernel
    if (solvers & (1u << BA_SOLVER_BAND)) hipLaunchKernelGGL(k_ldlt_band, dim3(1, nwin), dim3(LD_THREADS), bak_ldlt_smem(Npad), st, wins);
    if (solvers & (1u << BA_SOLVER_BLOCKED)) hipLaunchKernelGGL(k_ldlt_blocked, dim3(1, nwin), dim3(LD_THREADS), bak_ldlt_smem(Npad), st, wins);
    if (ev) (void)hipEventRecord(ev[3], st);
    const int nb = (max_pt * BA_PG > max_kf ? max_pt * BA_PG : max_kf);
    hipLaunchKernelGGL(k_backsub_update, dim3((nb + 255) / 256, nwin), dim3(256), 0, st, wins);
    if (ev) (void)hipEventRecord(ev[4], st);
    hipLaunchKernelGGL(k_errors, ge, dim3(256), 0, st, wins);
    hipLaunchKernelGGL(k_decide, dim3(1, nwin), dim3(256), 0, st, wins);
    if (ev) (void)hipEventRecord(ev[5], st);
}

void bak_final(hipStream_t st, BaWin* wins, const BaIo* io, int nwin, int max_kf, int max_pt, int max_edge) {
    const int nb = std::max(std::max(max_edge, 3 * max_pt), std::max(max_kf, (int)(sizeof(BaState) / 8)));
    hipLaunchKernelGGL(k_final, dim3((nb + 255) / 256, nwin), dim3(256), 0, st, wins, io);
}
