// bow.hip — vocabulary-node search (include/slamit.h, slamit_bow_search).
//
// Reference: ORB_SLAM2/src/ORBmatcher.cc:161-290 (SearchByBoW KeyFrame/Frame), :526-657 (SearchByBoW KeyFrame/KeyFrame),
// :659-826 (SearchForTriangulation), :141-158 (CheckDistEpipolarLine).
//
// One WAVEFRONT per group (= vocabulary node present on both sides).  A group's queries are walked in order, because in
// mode 0 an accepted query removes its candidate from the ones that follow; lanes hold the group's candidates (lane l:
// list positions l, l + 64, ...; a lane's matched positions are bits of one register).  "best / second best with strict
// '<', first one wins" is "the two smallest keys (distance << 16 | position)"; mode 1's "minimum distance, last one wins"
// is the smallest key (distance << 16 | 0xFFFF - position).  Wave-wide minima are DPP reductions.
// Float expressions are written exactly as the reference writes them; compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/slamit.h"
#include "slamit_internal.h"

struct BowDev {
    int n_groups, n1;
    const int* q_ptr; const int* q_idx; const int* c_ptr; const int* c_idx;
    const uint8_t* desc1; const uint8_t* desc2; const uint8_t* valid1; const uint8_t* valid2;   // valid* may be null
    int mode, th, th_inclusive; float nnratio;
    float F[9]; float ex, ey;
    const float* kp1; const float* kp2; const int* oct2;
    float scale[16], sigma2[16];
    int* match12; int* dist12; int* nmatches;
};

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
// wave-wide minimum as a wave-uniform value (four DPP exchanges inside the 16-lane rows, four v_readlane across them)
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
    v = min(v, dpp_u32<0xB1>(v));     // quad_perm [1,0,3,2]
    v = min(v, dpp_u32<0x4E>(v));     // quad_perm [2,3,0,1]
    v = min(v, dpp_u32<0x141>(v));    // row_half_mirror
    v = min(v, dpp_u32<0x140>(v));    // row_mirror
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}

__global__ __launch_bounds__(256) void bow_init_kernel(BowDev D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) *D.nmatches = 0;
    if (i < D.n1) { D.match12[i] = -1; D.dist12[i] = 256; }
}

__global__ __launch_bounds__(64) void bow_search_kernel(BowDev D) {
    const int g = blockIdx.x, lane = threadIdx.x;
    const int q0 = D.q_ptr[g], q1 = D.q_ptr[g + 1], c0 = D.c_ptr[g], nc = D.c_ptr[g + 1] - c0;
    if (nc <= 0 || q1 <= q0) return;
    const int nj = (nc + 63) >> 6;            // list positions per lane (<= 32: SLAMIT_BOW_MAX_GROUP)
    // positions this lane may not use: past the list, or not allowed by valid2
    unsigned blocked = 0;
    for (int j = 0; j < nj; ++j) {
        const int pos = lane + 64 * j;
        bool off = pos >= nc;
        if (!off && D.valid2) off = D.valid2[D.c_idx[c0 + pos]] == 0;
        blocked |= (unsigned)off << j;
    }
    int accepted = 0;
    for (int qi = q0; qi < q1; ++qi) {
        const int i1 = D.q_idx[qi];
        if (D.valid1 && !D.valid1[i1]) continue;   // wave-uniform
        const uint4* dq = reinterpret_cast<const uint4*>(D.desc1 + 32 * (size_t)i1);
        const uint4 a0 = dq[0], a1 = dq[1];
        float la = 0.f, lb = 0.f, lc = 0.f;       // epipolar line of the query in image 2 (mode 1)
        if (D.mode == 1) {
            const float x1 = D.kp1[2 * i1], y1 = D.kp1[2 * i1 + 1];
            la = x1 * D.F[0] + y1 * D.F[3] + D.F[6];
            lb = x1 * D.F[1] + y1 * D.F[4] + D.F[7];
            lc = x1 * D.F[2] + y1 * D.F[5] + D.F[8];
        }
        unsigned kb = 0xFFFFFFFFu, ks = 0xFFFFFFFFu;
        for (int j = 0; j < nj; ++j) {
            if ((blocked >> j) & 1u) continue;
            const int pos = lane + 64 * j;
            const int i2 = D.c_idx[c0 + pos];
            const uint4* dt = reinterpret_cast<const uint4*>(D.desc2 + 32 * (size_t)i2);
            const uint4 t0 = dt[0], t1 = dt[1];
            const unsigned d = __popc(a0.x ^ t0.x) + __popc(a0.y ^ t0.y) + __popc(a0.z ^ t0.z) + __popc(a0.w ^ t0.w) +
                               __popc(a1.x ^ t1.x) + __popc(a1.y ^ t1.y) + __popc(a1.z ^ t1.z) + __popc(a1.w ^ t1.w);
            if (D.mode == 0) {
                if (d >= 256u) continue;             // 'dist < bestDist1' can never hold against the initial 256
                const unsigned k = (d << 16) | (unsigned)pos;
                ks = min(ks, max(kb, k));
                kb = min(kb, k);
            } else {
                if ((int)d > D.th) continue;          // :731 (bestDist starts at TH_LOW; the running bound is the wave minimum)
                const float x2 = D.kp2[2 * i2], y2 = D.kp2[2 * i2 + 1];
                const int oc = D.oct2[i2] & 15;
                const float distex = D.ex - x2, distey = D.ey - y2;
                if (distex * distex + distey * distey < 100 * D.scale[oc]) continue;   // :741
                const float num = la * x2 + lb * y2 + lc;
                const float den = la * la + lb * lb;
                if (den == 0) continue;
                const float dsqr = num * num / den;
                if (!((double)dsqr < 3.84 * (double)D.sigma2[oc])) continue;           // :157 (double comparison)
                kb = min(kb, (d << 16) | (0xFFFFu - (unsigned)pos));
            }
        }
        const unsigned best = wave_min_u32(kb);
        if (best == 0xFFFFFFFFu) continue;            // no candidate at all: dist12 stays 256
        const int bd = (int)(best >> 16);
        if (D.mode == 0) {
            const unsigned second = wave_min_u32(kb == best ? ks : kb);   // keys are unique: one lane holds the best
            const int sd = second == 0xFFFFFFFFu ? 256 : (int)(second >> 16);
            const bool ok = (D.th_inclusive ? bd <= D.th : bd < D.th) && (float)bd < D.nnratio * (float)sd;
            const int pos = (int)(best & 0xFFFFu);
            if (ok && lane == (pos & 63)) blocked |= 1u << (pos >> 6);
            if (lane == 0) { D.dist12[i1] = bd; if (ok) D.match12[i1] = D.c_idx[c0 + pos]; }
            accepted += ok;
        } else {
            const int pos = 0xFFFF - (int)(best & 0xFFFFu);
            if (lane == 0) { D.dist12[i1] = bd; D.match12[i1] = D.c_idx[c0 + pos]; }
            ++accepted;
        }
    }
    if (lane == 0 && accepted) atomicAdd(D.nmatches, accepted);
}

extern "C" int slamit_bow_search(int device, const uint8_t* desc1, int32_t n1, const uint8_t* valid1, const uint8_t* desc2,
                                 int32_t n2, const uint8_t* valid2, const slamit_bow_groups* G, const slamit_bow_rule* rule,
                                 int32_t* match12, int32_t* dist12, int32_t* nmatches) {
    if (!G || !rule || !nmatches || n1 < 0 || n2 < 0 || G->n_groups < 0 || (rule->mode != 0 && rule->mode != 1))
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_bow_search: bad argument");
    *nmatches = 0;
    if (n1 == 0) return SLAMIT_OK;
    if (!match12 || !desc1 || (n2 && !desc2)) return slamit_fail(SLAMIT_ERR_ARG, "slamit_bow_search: null array");
    for (int i = 0; i < n1; ++i) { match12[i] = -1; if (dist12) dist12[i] = 256; }
    const int ng = G->n_groups;
    if (ng == 0 || n2 == 0) return SLAMIT_OK;
    if (!G->q_ptr || !G->q_idx || !G->c_ptr || !G->c_idx) return slamit_fail(SLAMIT_ERR_ARG, "slamit_bow_search: null group arrays");
    if (rule->mode == 1 && (!rule->kp1_xy || !rule->kp2_xy || !rule->kp2_octave))
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_bow_search: mode 1 needs kp1_xy, kp2_xy, kp2_octave");
    const int nq = G->q_ptr[ng], ncand = G->c_ptr[ng];
    if (G->q_ptr[0] != 0 || G->c_ptr[0] != 0 || nq < 0 || ncand < 0) return slamit_fail(SLAMIT_ERR_ARG, "slamit_bow_search: malformed group offsets");
    {   // a feature sits in one vocabulary node: an index may appear once per side (this is what makes groups independent)
        std::vector<uint8_t> seen1((size_t)n1, 0), seen2((size_t)n2, 0);
        for (int g = 0; g < ng; ++g) {
            if (G->q_ptr[g + 1] < G->q_ptr[g] || G->c_ptr[g + 1] < G->c_ptr[g]) return slamit_fail(SLAMIT_ERR_ARG, "slamit_bow_search: malformed group offsets");
            if (G->c_ptr[g + 1] - G->c_ptr[g] > SLAMIT_BOW_MAX_GROUP) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_bow_search: a group holds more than SLAMIT_BOW_MAX_GROUP candidates");
        }
        for (int k = 0; k < nq; ++k) {
            const int i = G->q_idx[k];
            if (i < 0 || i >= n1 || seen1[i]) return slamit_fail(SLAMIT_ERR_ARG, "slamit_bow_search: side-1 index out of range or repeated");
            seen1[i] = 1;
        }
        for (int k = 0; k < ncand; ++k) {
            const int i = G->c_idx[k];
            if (i < 0 || i >= n2 || seen2[i]) return slamit_fail(SLAMIT_ERR_ARG, "slamit_bow_search: side-2 index out of range or repeated");
            seen2[i] = 1;
        }
    }
    if (nq == 0 || ncand == 0) return SLAMIT_OK;
    SLAMIT_USE_DEVICE(device);
    // one pinned staging block and one device slab per host thread, kept between calls (LocalMapping makes this call for
    // every neighbour keyframe of every new keyframe); inputs first, outputs last
    const bool m1 = rule->mode == 1;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_d1 = take(32 * (size_t)n1), o_d2 = take(32 * (size_t)n2), o_v1 = take((size_t)n1), o_v2 = take((size_t)n2);
    const size_t o_qp = take(4 * (size_t)(ng + 1)), o_cp = take(4 * (size_t)(ng + 1)), o_qi = take(4 * (size_t)nq), o_ci = take(4 * (size_t)ncand);
    const size_t o_k1 = take(m1 ? 8 * (size_t)n1 : 0), o_k2 = take(m1 ? 8 * (size_t)n2 : 0), o_oc = take(m1 ? 4 * (size_t)n2 : 0);
    const size_t in_bytes = off;
    const size_t o_m = take(4 * (size_t)n1), o_d = take(4 * (size_t)n1), o_nm = take(4);
    const size_t io_bytes = off;
    static thread_local SlamitScratch S;
    {
        const hipError_t es = slamit_scratch_reserve(S, device, io_bytes);
        if (es != hipSuccess) return slamit_fail_hip(es, "slamit_bow_search: scratch");
    }
    uint8_t* hb = S.host;
    uint8_t* d = S.dev;
    memcpy(hb + o_d1, desc1, 32 * (size_t)n1); memcpy(hb + o_d2, desc2, 32 * (size_t)n2);
    if (valid1) memcpy(hb + o_v1, valid1, (size_t)n1);
    if (valid2) memcpy(hb + o_v2, valid2, (size_t)n2);
    memcpy(hb + o_qp, G->q_ptr, 4 * (size_t)(ng + 1)); memcpy(hb + o_cp, G->c_ptr, 4 * (size_t)(ng + 1));
    memcpy(hb + o_qi, G->q_idx, 4 * (size_t)nq); memcpy(hb + o_ci, G->c_idx, 4 * (size_t)ncand);
    if (m1) { memcpy(hb + o_k1, rule->kp1_xy, 8 * (size_t)n1); memcpy(hb + o_k2, rule->kp2_xy, 8 * (size_t)n2); memcpy(hb + o_oc, rule->kp2_octave, 4 * (size_t)n2); }
    hipError_t e = hipMemcpyAsync(d, hb, in_bytes, hipMemcpyHostToDevice, S.st);
    if (e == hipSuccess) {
        BowDev D;
        D.n_groups = ng; D.n1 = n1;
        D.q_ptr = (const int*)(d + o_qp); D.q_idx = (const int*)(d + o_qi); D.c_ptr = (const int*)(d + o_cp); D.c_idx = (const int*)(d + o_ci);
        D.desc1 = d + o_d1; D.desc2 = d + o_d2; D.valid1 = valid1 ? d + o_v1 : nullptr; D.valid2 = valid2 ? d + o_v2 : nullptr;
        D.mode = rule->mode; D.th = rule->th; D.th_inclusive = rule->th_inclusive; D.nnratio = rule->nnratio;
        memcpy(D.F, rule->F12, sizeof(D.F)); D.ex = rule->ex; D.ey = rule->ey;
        D.kp1 = (const float*)(d + o_k1); D.kp2 = (const float*)(d + o_k2); D.oct2 = (const int*)(d + o_oc);
        memcpy(D.scale, rule->scale_factor, sizeof(D.scale)); memcpy(D.sigma2, rule->level_sigma2, sizeof(D.sigma2));
        D.match12 = (int*)(d + o_m); D.dist12 = (int*)(d + o_d); D.nmatches = (int*)(d + o_nm);
        hipLaunchKernelGGL(bow_init_kernel, dim3((n1 + 255) / 256), dim3(256), 0, S.st, D);
        hipLaunchKernelGGL(bow_search_kernel, dim3(ng), dim3(64), 0, S.st, D);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hb + in_bytes, d + in_bytes, io_bytes - in_bytes, hipMemcpyDeviceToHost, S.st);
    if (e == hipSuccess) e = hipStreamSynchronize(S.st);
    if (e != hipSuccess) return slamit_fail_hip(e, "slamit_bow_search");
    memcpy(match12, hb + o_m, 4 * (size_t)n1);
    if (dist12) memcpy(dist12, hb + o_d, 4 * (size_t)n1);
    memcpy(nmatches, hb + o_nm, 4);
    return SLAMIT_OK;
}
