// se3_device.h — fp64 SE3 / quaternion helpers shared by the bundle-adjustment and pose-optimisation
// kernels.  Same closed forms as Eigen's Quaterniond and g2o's SE3Quat (Thirdparty/g2o/g2o/types/se3quat.h).
#ifndef SLAMIT_SE3_DEVICE_H
#define SLAMIT_SE3_DEVICE_H
#include <hip/hip_runtime.h>

// ---- small fp64 helpers (same formulas as Eigen / g2o) ---------------------------------------
__device__ __forceinline__ void quat_rot(const double* q, const double* v, double* r) {
    double uvx = 2 * (q[1] * v[2] - q[2] * v[1]), uvy = 2 * (q[2] * v[0] - q[0] * v[2]), uvz = 2 * (q[0] * v[1] - q[1] * v[0]);
    r[0] = v[0] + q[3] * uvx + (q[1] * uvz - q[2] * uvy);
    r[1] = v[1] + q[3] * uvy + (q[2] * uvx - q[0] * uvz);
    r[2] = v[2] + q[3] * uvz + (q[0] * uvy - q[1] * uvx);
}
__device__ __forceinline__ void quat_to_R(const double* q, double* R) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// One case of Eigen's matrix -> quaternion branch for a negative trace, largest diagonal element I.  The indices are
// compile-time constants: run-time indices into m[] would put the caller's rotation matrix into scratch memory.
template <int I>
__device__ __forceinline__ void R_to_quat_case(const double* m, double* q) {
    constexpr int J = (I + 1) % 3, K = (J + 1) % 3;
    double t = sqrt(m[4 * I] - m[4 * J] - m[4 * K] + 1.0);
    q[I] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[3 * K + J] - m[3 * J + K]) * t;
    q[J] = (m[3 * J + I] + m[3 * I + J]) * t;
    q[K] = (m[3 * K + I] + m[3 * I + K]) * t;
}
__device__ __forceinline__ void R_to_quat(const double* m, double* q) {
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > (i ? m[4] : m[0])) i = 2;
        if (i == 0) R_to_quat_case<0>(m, q);
        else if (i == 1) R_to_quat_case<1>(m, q);
        else R_to_quat_case<2>(m, q);
    }
}
__device__ __forceinline__ void quat_normalize(double* q) {
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
// T <- exp(u) * T   (SE3Quat::exp, g2o/types/se3quat.h:218-253; u = [omega, upsilon])
__device__ void pose_oplus(double* T, const double* u) {
    const double wx = u[0], wy = u[1], wz = u[2];
    const double theta = sqrt(wx * wx + wy * wy + wz * wz);
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) O2[3 * i + j] = O[3 * i] * O[j] + O[3 * i + 1] * O[3 + j] + O[3 * i + 2] * O[6 + j];
    double R[9], V[9];
    if (theta < 0.00001) {
        for (int i = 0; i < 9; ++i) { R[i] = (i % 4 == 0 ? 1.0 : 0.0) + O[i] + O2[i]; V[i] = R[i]; }
    } else {
        const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / (theta * theta * theta);
        for (int i = 0; i < 9; ++i) {
            const double I = (i % 4 == 0 ? 1.0 : 0.0);
            R[i] = I + a * O[i] + b * O2[i];
            V[i] = I + b * O[i] + c * O2[i];
        }
    }
    double qe[4], te[3], rt[3], nq[4];
    R_to_quat(R, qe);
    quat_normalize(qe);
    for (int i = 0; i < 3; ++i) te[i] = V[3 * i] * u[3] + V[3 * i + 1] * u[4] + V[3 * i + 2] * u[5];
    quat_rot(qe, T + 4, rt);
    const double* b4 = T;
    nq[3] = qe[3] * b4[3] - qe[0] * b4[0] - qe[1] * b4[1] - qe[2] * b4[2];
    nq[0] = qe[3] * b4[0] + qe[0] * b4[3] + qe[1] * b4[2] - qe[2] * b4[1];
    nq[1] = qe[3] * b4[1] + qe[1] * b4[3] + qe[2] * b4[0] - qe[0] * b4[2];
    nq[2] = qe[3] * b4[2] + qe[2] * b4[3] + qe[0] * b4[1] - qe[1] * b4[0];
    quat_normalize(nq);
    T[0] = nq[0]; T[1] = nq[1]; T[2] = nq[2]; T[3] = nq[3];
    T[4] = te[0] + rt[0]; T[5] = te[1] + rt[1]; T[6] = te[2] + rt[2];
}

// value of lane `l` (compile-time constant) as a wave-uniform scalar: v_readlane_b32 x2, no LDS
__device__ __forceinline__ double readlane_d(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// quad_perm exchange in which every lane has a source (CTRL < 0x100): no zero-initialised "old" operand to set up
template <int CTRL>
__device__ __forceinline__ double dpp_quad_d(double v) {
    const int l = __double2loint(v), h = __double2hiint(v);
    const int lo = __builtin_amdgcn_update_dpp(l, l, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(h, h, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// value of lane `l` (wave-uniform, not a constant) as a wave-uniform scalar
__device__ __forceinline__ double readlane_dyn_d(double v, int l) {
    const int ls = __builtin_amdgcn_readfirstlane(l);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), ls);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), ls);
    return __hiloint2double(hi, lo);
}

// Sum over the 64 lanes, the same value in every lane, in a fixed order: four DPP exchanges make every 16-lane row hold its
// row sum (xor 1, xor 2, half-row mirror, row mirror), four v_readlane pairs join the rows.  (The ds_bpermute form of
// __shfl_xor is an LDS round trip per step: twelve of them per sum, and the optimisers reduce 27-35 sums per iteration.)
template <int CTRL>
__device__ __forceinline__ double dpp_xchg_d(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_xchg_d<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_xchg_d<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_xchg_d<0x141>(v);   // row_half_mirror
    v += dpp_xchg_d<0x140>(v);   // row_mirror
    return ((readlane_d(v, 0) + readlane_d(v, 16)) + readlane_d(v, 32)) + readlane_d(v, 48);
}

#endif
