// search.hip — guided search: Frame::GetFeaturesInArea + Hamming best/second + greedy take
// (include/slamit.h, slamit_guided_search).
//
// Reference: ORB_SLAM2/src/ORBmatcher.cc:47-131 and :1332-1474 (the per-query loop bodies),
// ORB_SLAM2/src/Frame.cc:336-357, 447-517 (grid assignment and window query).
//
// Two kernels:
//   search_candidates_kernel  one wavefront per query, all queries in parallel: every keypoint is tested
//       against the window exactly as GetFeaturesInArea does (grid cell range from the reference's float
//       expressions, level range, |dx| < r && |dy| < r); hits are appended to the query's candidate list
//       as one u64 key  (distance << 32) | (cell_x * 48 + cell_y) << 13 | keypoint  — sorting by that key IS
//       the reference's scan order (cells x-major, then y, then insertion = keypoint index) with the
//       Hamming distance in front, and "best / second best with strict <" equals "two smallest keys".
//   search_resolve_kernel     one wavefront walks the queries IN ORDER (the reference marks the winning
//       keypoint as taken before it looks at the next map point): lanes hold the candidates, keypoints
//       already taken are masked out, two wave-wide min-reductions give best and second, the acceptance
//       rule runs, the taken bit is set in LDS.
// Float expressions are written exactly as the reference writes them; compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/slamit.h"
#include "slamit_internal.h"

#define GRID_COLS 64   // FRAME_GRID_COLS, include/Frame.h:41
#define GRID_ROWS 48   // FRAME_GRID_ROWS, include/Frame.h:40

struct SearchDev {
    int n, m, cand_cap;
    const float* kp_xy; const int* kp_octave; const uint8_t* kp_desc; const uint8_t* kp_taken;
    float min_x, min_y, inv_w, inv_h;
    const float* uvr; const int* lmin; const int* lmax; const uint8_t* qdesc; const uint8_t* valid; const uint8_t* takes;
    unsigned long long* cand;   // m x cand_cap
    int* cand_n;                // m (may exceed cand_cap: overflow flag)
    int th_dist, use_ratio; float nnratio;
    int* match_kp; int* out4;   // out4: m x 4 (best_dist, best_level, second_dist, second_level)
    int* nmatches;
};

__global__ __launch_bounds__(256) void search_candidates_kernel(SearchDev D) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= D.m) return;
    if (lane == 0) D.cand_n[q] = 0;
    if (!D.valid[q]) return;
    const float x = D.uvr[3 * q], y = D.uvr[3 * q + 1], r = D.uvr[3 * q + 2];
    // Frame::GetFeaturesInArea, Frame.cc:452-466
    const int nMinCellX = max(0, (int)floorf((x - D.min_x - r) * D.inv_w));
    if (nMinCellX >= GRID_COLS) return;
    const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf((x - D.min_x + r) * D.inv_w));
    if (nMaxCellX < 0) return;
    const int nMinCellY = max(0, (int)floorf((y - D.min_y - r) * D.inv_h));
    if (nMinCellY >= GRID_ROWS) return;
    const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf((y - D.min_y + r) * D.inv_h));
    if (nMaxCellY < 0) return;
    const int minLevel = D.lmin[q], maxLevel = D.lmax[q];
    const uint4* Q = reinterpret_cast<const uint4*>(D.qdesc + 32 * (size_t)q);
    const uint4 a0 = Q[0], a1 = Q[1];
    unsigned long long* out = D.cand + (size_t)q * D.cand_cap;
    int count = 0;
    for (int i0 = 0; i0 < D.n; i0 += 64) {
        const int i = i0 + lane;
        bool hit = false;
        unsigned long long key = 0;
        if (i < D.n) {
            const float px = D.kp_xy[2 * i], py = D.kp_xy[2 * i + 1];
            // Frame::PosInGrid, Frame.cc:505-517 (round = half away from zero)
            const int posX = (int)roundf((px - D.min_x) * D.inv_w), posY = (int)roundf((py - D.min_y) * D.inv_h);
            const bool ingrid = !(posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS);
            const int oct = D.kp_octave[i];
            const bool lev = !(oct < minLevel) && !(maxLevel >= 0 && oct > maxLevel);
            const float distx = px - x, disty = py - y;
            hit = ingrid && posX >= nMinCellX && posX <= nMaxCellX && posY >= nMinCellY && posY <= nMaxCellY && lev &&
                  fabsf(distx) < r && fabsf(disty) < r;
            if (hit) {
                const uint4* T = reinterpret_cast<const uint4*>(D.kp_desc + 32 * (size_t)i);
                const uint4 t0 = T[0], t1 = T[1];
                const int d = __popc(a0.x ^ t0.x) + __popc(a0.y ^ t0.y) + __popc(a0.z ^ t0.z) + __popc(a0.w ^ t0.w) +
                              __popc(a1.x ^ t1.x) + __popc(a1.y ^ t1.y) + __popc(a1.z ^ t1.z) + __popc(a1.w ^ t1.w);
                hit = d < 256;   // bestDist starts at 256 and the test is a strict '<': a complement never wins
                key = ((unsigned long long)d << 32) | ((unsigned long long)(posX * GRID_ROWS + posY) << 13) | (unsigned)i;
            }
        }
        const unsigned long long m = __ballot(hit);
        if (hit) {
            const int o = count + __popcll(m & ((1ull << lane) - 1ull));
            if (o < D.cand_cap) out[o] = key;
        }
        count += __popcll(m);
    }
    if (lane == 0) D.cand_n[q] = count;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, d, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), d, 64);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

__global__ __launch_bounds__(64) void search_resolve_kernel(SearchDev D) {
    __shared__ unsigned taken[(SLAMIT_SEARCH_MAX_KP + 32) / 32];
    const int lane = threadIdx.x;
    for (int w = lane; w < (D.n + 31) / 32; w += 64) {
        unsigned bits = 0;
        for (int b = 0; b < 32; ++b) { const int i = 32 * w + b; if (i < D.n && D.kp_taken[i]) bits |= 1u << b; }
        taken[w] = bits;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long NONE = ~0ull;
    int nmatches = 0;
    for (int q = 0; q < D.m; ++q) {
        const int nc = min(D.cand_n[q], D.cand_cap);
        int res = -1, bd = 256, bl = -1, sd = 256, sl = -1;
        if (nc > 0) {
            const unsigned long long* C = D.cand + (size_t)q * D.cand_cap;
            unsigned long long k1 = NONE, k2 = NONE;   // this lane's two smallest live keys
            for (int c = lane; c < nc; c += 64) {
                const unsigned long long k = C[c];
                const int i = (int)(k & 8191);
                if ((taken[i >> 5] >> (i & 31)) & 1u) continue;   // F.mvpMapPoints[idx] with observations
                if (k < k1) { k2 = k1; k1 = k; } else if (k < k2) k2 = k;
            }
            const unsigned long long best = wave_min_u64(k1);
            // second = smallest key other than `best` (keys are unique: they carry the keypoint index)
            const unsigned long long mine2 = (k1 == best) ? k2 : k1;
            const unsigned long long second = wave_min_u64(mine2);
            if (best != NONE) {
                const int bi = (int)(best & 8191);
                bd = (int)(best >> 32); bl = D.kp_octave[bi];
                if (second != NONE) { sd = (int)(second >> 32); sl = D.kp_octave[(int)(second & 8191)]; }
                if (bd <= D.th_dist) {   // ORBmatcher.cc:120-128
                    const bool reject = D.use_ratio && bl == sl && (float)bd > D.nnratio * (float)sd;
                    if (!reject) {
                        res = bi;
                        if (lane == 0 && D.takes[q]) taken[bi >> 5] |= 1u << (bi & 31);
                        ++nmatches;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) {
            D.match_kp[q] = res;
            D.out4[4 * q] = bd; D.out4[4 * q + 1] = bl; D.out4[4 * q + 2] = sd; D.out4[4 * q + 3] = sl;
        }
    }
    if (lane == 0) *D.nmatches = nmatches;
}

extern "C" int slamit_guided_search(int device, const slamit_frame_view* F, const slamit_search_queries* Q,
                                    const slamit_search_rule* rule, int32_t* match_kp, int32_t* nmatches, int32_t* best_dist,
                                    int32_t* best_level, int32_t* second_dist, int32_t* second_level) {
    if (!F || !Q || !rule || !nmatches || F->n < 0 || Q->m < 0) return slamit_fail(SLAMIT_ERR_ARG, "slamit_guided_search: bad argument");
    *nmatches = 0;
    if (Q->m == 0) return SLAMIT_OK;
    if (!match_kp || !Q->uvr || !Q->level_min || !Q->level_max || !Q->desc || !Q->valid ||
        (F->n && (!F->kp_xy || !F->kp_octave || !F->desc || !F->kp_taken)))
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_guided_search: null array");
    if (F->n > SLAMIT_SEARCH_MAX_KP) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_guided_search: more than SLAMIT_SEARCH_MAX_KP keypoints");
    HIP_TRY(hipSetDevice(device));
    const int n = F->n, m = Q->m, cap = std::min(std::max(n, 1), SLAMIT_SEARCH_MAX_CAND);
    // one slab: keypoints | queries | candidate lists | outputs
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_xy = take(8 * (size_t)n), o_oct = take(4 * (size_t)n), o_kd = take(32 * (size_t)n), o_tk = take((size_t)n);
    const size_t o_uvr = take(12 * (size_t)m), o_l0 = take(4 * (size_t)m), o_l1 = take(4 * (size_t)m), o_qd = take(32 * (size_t)m), o_va = take((size_t)m), o_tq = take((size_t)m);
    const size_t o_cand = take(8 * (size_t)m * cap), o_cn = take(4 * (size_t)m), o_mk = take(4 * (size_t)m), o_o4 = take(16 * (size_t)m), o_nm = take(4);
    uint8_t* d = nullptr;
    hipError_t e = hipMalloc((void**)&d, off);
#define UP(o, src, bytes) if (e == hipSuccess && (bytes)) e = hipMemcpy(d + (o), (src), (bytes), hipMemcpyHostToDevice)
    UP(o_xy, F->kp_xy, 8 * (size_t)n); UP(o_oct, F->kp_octave, 4 * (size_t)n); UP(o_kd, F->desc, 32 * (size_t)n); UP(o_tk, F->kp_taken, (size_t)n);
    UP(o_uvr, Q->uvr, 12 * (size_t)m); UP(o_l0, Q->level_min, 4 * (size_t)m); UP(o_l1, Q->level_max, 4 * (size_t)m);
    UP(o_qd, Q->desc, 32 * (size_t)m); UP(o_va, Q->valid, (size_t)m);
    if (Q->takes) { UP(o_tq, Q->takes, (size_t)m); } else if (e == hipSuccess) e = hipMemset(d + o_tq, 1, (size_t)m);
#undef UP
    std::vector<int> cn(m), mk(m), o4(4 * (size_t)m);
    int nm = 0;
    if (e == hipSuccess) {
        SearchDev D;
        D.n = n; D.m = m; D.cand_cap = cap;
        D.kp_xy = (const float*)(d + o_xy); D.kp_octave = (const int*)(d + o_oct); D.kp_desc = d + o_kd; D.kp_taken = d + o_tk;
        D.min_x = F->min_x; D.min_y = F->min_y; D.inv_w = F->inv_w; D.inv_h = F->inv_h;
        D.uvr = (const float*)(d + o_uvr); D.lmin = (const int*)(d + o_l0); D.lmax = (const int*)(d + o_l1); D.qdesc = d + o_qd; D.valid = d + o_va; D.takes = d + o_tq;
        D.cand = (unsigned long long*)(d + o_cand); D.cand_n = (int*)(d + o_cn);
        D.th_dist = rule->th_dist; D.use_ratio = rule->use_ratio; D.nnratio = rule->nnratio;
        D.match_kp = (int*)(d + o_mk); D.out4 = (int*)(d + o_o4); D.nmatches = (int*)(d + o_nm);
        hipLaunchKernelGGL(search_candidates_kernel, dim3((m + 3) / 4), dim3(256), 0, 0, D);
        hipLaunchKernelGGL(search_resolve_kernel, dim3(1), dim3(64), 0, 0, D);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(cn.data(), d + o_cn, 4 * (size_t)m, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(mk.data(), d + o_mk, 4 * (size_t)m, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(o4.data(), d + o_o4, 16 * (size_t)m, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(&nm, d + o_nm, 4, hipMemcpyDeviceToHost);
    hipFree(d);
    if (e != hipSuccess) return slamit_fail_hip(e, "slamit_guided_search");
    for (int q = 0; q < m; ++q)
        if (cn[q] > cap) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_guided_search: a window holds more than SLAMIT_SEARCH_MAX_CAND keypoints");
    memcpy(match_kp, mk.data(), 4 * (size_t)m);
    *nmatches = nm;
    for (int q = 0; q < m; ++q) {
        if (best_dist) best_dist[q] = o4[4 * (size_t)q];
        if (best_level) best_level[q] = o4[4 * (size_t)q + 1];
        if (second_dist) second_dist[q] = o4[4 * (size_t)q + 2];
        if (second_level) second_level[q] = o4[4 * (size_t)q + 3];
    }
    return SLAMIT_OK;
}
