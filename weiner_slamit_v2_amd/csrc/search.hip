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

// One frame of a batch: blockIdx.y (candidates) / blockIdx.x (resolve) selects it; all arrays are strided per frame.
struct SearchDev {
    int nframes, kp_cap, q_cap, cand_cap;
    const int* n_arr; int n_fixed;           // keypoints per frame
    const int* m_arr; int m_fixed;           // queries per frame
    const uint8_t* kp; int kp_rec;           // keypoint records: float x, y at byte 0 / 4, int octave at kp_oct_off; kp_rec bytes each
    int kp_oct_off;
    const uint8_t* kp_desc; const uint8_t* kp_taken;
    float min_x, min_y, inv_w, inv_h;
    const float* uvr; const int* lmin; const int* lmax; const uint8_t* qdesc; const uint8_t* valid; const uint8_t* takes;
    unsigned long long* cand;   // [nframes][q_cap][cand_cap]
    int* cand_n;                // [nframes][q_cap] (may exceed cand_cap: overflow)
    unsigned long long* tent;   // [nframes][q_cap][2]: the two smallest keys among the candidates not taken ON ENTRY
    int th_dist, use_ratio; float nnratio;
    float chi2_gate; float inv_sigma2[16];   // Fuse's reprojection gate (chi2_gate <= 0: off)
    int mode;                                 // 0 taken flags, 1 SearchForInitialization's matched-distance state
    int* match_kp; int* out4;   // [nframes][q_cap], [nframes][q_cap][4] (best_dist, best_level, second_dist, second_level) or null
    int* nmatches;              // [nframes]
};

// key = distance << 32 | cell << 17 | keypoint << 4 | octave: ordered by (distance, cell, keypoint) = the reference's
// scan order; the octave rides along in the low bits so the resolve step never goes back to the keypoint table
#define KEY_KP(k) ((int)(((k) >> 4) & 8191))
#define KEY_OCT(k) ((int)((k) & 15))

// wave-wide minimum of a u64 as a wave-uniform value: four DPP exchanges make every 16-lane row uniform (xor 1, xor 2,
// half-row mirror, row mirror), four v_readlane + scalar mins join the rows.  (The ds_bpermute form of __shfl_xor costs
// an LDS round trip per step: 12 of them per reduction were the whole microsecond this serial loop spent per query.)
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, false);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l) << 32) |
           (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
}
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    unsigned long long o;
    o = dpp_u64<0xB1>(v); v = o < v ? o : v;     // quad_perm [1,0,3,2]
    o = dpp_u64<0x4E>(v); v = o < v ? o : v;     // quad_perm [2,3,0,1]
    o = dpp_u64<0x141>(v); v = o < v ? o : v;    // row_half_mirror
    o = dpp_u64<0x140>(v); v = o < v ? o : v;    // row_mirror
    const unsigned long long a = readlane_u64(v, 0), b = readlane_u64(v, 16), c = readlane_u64(v, 32), d = readlane_u64(v, 48);
    const unsigned long long ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
}

// A query's search window (Frame::GetFeaturesInArea, Frame.cc:452-466) and descriptor; ok = the window meets the grid.
struct QueryWin {
    float x, y, r;
    int nMinCellX, nMaxCellX, nMinCellY, nMaxCellY, minLevel, maxLevel;
    uint4 a0, a1;
};
__device__ __forceinline__ bool query_window(const SearchDev& D, size_t qo, QueryWin& W) {
    W.x = D.uvr[3 * qo]; W.y = D.uvr[3 * qo + 1]; W.r = D.uvr[3 * qo + 2];
    W.nMinCellX = max(0, (int)floorf((W.x - D.min_x - W.r) * D.inv_w));
    if (W.nMinCellX >= GRID_COLS) return false;
    W.nMaxCellX = min(GRID_COLS - 1, (int)ceilf((W.x - D.min_x + W.r) * D.inv_w));
    if (W.nMaxCellX < 0) return false;
    W.nMinCellY = max(0, (int)floorf((W.y - D.min_y - W.r) * D.inv_h));
    if (W.nMinCellY >= GRID_ROWS) return false;
    W.nMaxCellY = min(GRID_ROWS - 1, (int)ceilf((W.y - D.min_y + W.r) * D.inv_h));
    if (W.nMaxCellY < 0) return false;
    W.minLevel = D.lmin[qo]; W.maxLevel = D.lmax[qo];
    const uint4* Q = reinterpret_cast<const uint4*>(D.qdesc + 32 * qo);
    W.a0 = Q[0]; W.a1 = Q[1];
    return true;
}
// keypoint i against the window: is it a candidate, and its key (distance, cell, keypoint, octave)
__device__ __forceinline__ bool window_key(const SearchDev& D, const QueryWin& W, const uint8_t* KP, const uint8_t* KD, int i, unsigned long long& key) {
    const uint8_t* rec = KP + (size_t)i * D.kp_rec;
    const float px = *reinterpret_cast<const float*>(rec), py = *reinterpret_cast<const float*>(rec + 4);
    // Frame::PosInGrid, Frame.cc:505-517 (round = half away from zero)
    const int posX = (int)roundf((px - D.min_x) * D.inv_w), posY = (int)roundf((py - D.min_y) * D.inv_h);
    const bool ingrid = !(posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS);
    const int oct = *reinterpret_cast<const int*>(rec + D.kp_oct_off);
    const bool lev = !(oct < W.minLevel) && !(W.maxLevel >= 0 && oct > W.maxLevel);
    const float distx = px - W.x, disty = py - W.y;
    bool hit = ingrid && posX >= W.nMinCellX && posX <= W.nMaxCellX && posY >= W.nMinCellY && posY <= W.nMaxCellY && lev &&
               fabsf(distx) < W.r && fabsf(disty) < W.r;
    if (hit && D.chi2_gate > 0.f) {   // ORBmatcher::Fuse, ORBmatcher.cc:925-936 (mono branch)
        const float e2 = distx * distx + disty * disty;
        if (e2 * D.inv_sigma2[oct & 15] > D.chi2_gate) hit = false;
    }
    if (hit) {
        const uint4* T = reinterpret_cast<const uint4*>(KD + 32 * (size_t)i);
        const uint4 t0 = T[0], t1 = T[1];
        const int d = __popc(W.a0.x ^ t0.x) + __popc(W.a0.y ^ t0.y) + __popc(W.a0.z ^ t0.z) + __popc(W.a0.w ^ t0.w) +
                      __popc(W.a1.x ^ t1.x) + __popc(W.a1.y ^ t1.y) + __popc(W.a1.z ^ t1.z) + __popc(W.a1.w ^ t1.w);
        hit = d < 256;   // bestDist starts at 256 and the test is a strict '<': a complement never wins
        key = ((unsigned long long)d << 32) | ((unsigned long long)(posX * GRID_ROWS + posY) << 17) |
              ((unsigned long long)i << 4) | (unsigned long long)(oct & 15);
    }
    return hit;
}

__global__ __launch_bounds__(256) void search_candidates_kernel(SearchDev D) {
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.y;
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int m = D.m_arr ? min(D.m_arr[f], D.q_cap) : D.m_fixed;
    if (q >= m) return;
    const int n = D.n_arr ? min(D.n_arr[f], D.kp_cap) : D.n_fixed;
    const size_t qo = (size_t)f * D.q_cap + q;
    if (lane == 0) { D.cand_n[qo] = 0; D.tent[2 * qo] = ~0ull; D.tent[2 * qo + 1] = ~0ull; }
    if (!D.valid[qo]) return;
    QueryWin W;
    if (!query_window(D, qo, W)) return;
    unsigned long long* out = D.cand + qo * D.cand_cap;
    const uint8_t* KP = D.kp + (size_t)f * D.kp_cap * D.kp_rec;
    const uint8_t* KD = D.kp_desc + (size_t)f * D.kp_cap * 32;
    const uint8_t* TK = D.kp_taken + (size_t)f * D.kp_cap;
    int count = 0;
    unsigned long long k1 = ~0ull, k2 = ~0ull;   // this lane's two smallest keys among keypoints free on entry
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int i = i0 + lane;
        unsigned long long key = 0;
        const bool hit = i < n && window_key(D, W, KP, KD, i, key);
        const unsigned long long mk = __ballot(hit);
        if (hit) {
            const int o = count + __popcll(mk & ((1ull << lane) - 1ull));
            if (o < D.cand_cap) out[o] = key;
            if (D.mode == 1 || !TK[i]) { if (key < k1) { k2 = k1; k1 = key; } else if (key < k2) k2 = key; }
        }
        count += __popcll(mk);
    }
    // tentative (best, second): exact for the resolve pass unless an EARLIER query of this call takes one of the two
    const unsigned long long best = wave_min_u64(k1);
    const unsigned long long second = wave_min_u64(k1 == best ? k2 : k1);
    if (lane == 0) { D.cand_n[qo] = count; D.tent[2 * qo] = best; D.tent[2 * qo + 1] = second; }
}

// One wavefront per frame walks that frame's queries IN ORDER (the reference assigns the winning keypoint before it looks
// at the next map point).  The candidates kernel already found every query's two smallest keys among the keypoints free
// on entry; removing OTHER candidates cannot change the two smallest, so that pair is still the answer unless an earlier
// query of this call took one of the two.  The walk therefore only tests two taken bits per query (64 queries' pairs
// are loaded at once, lane j holding query j) and re-scans the candidate list on the rare conflict.
__global__ __launch_bounds__(64) void search_resolve_kernel(SearchDev D) {
    __shared__ unsigned taken[(SLAMIT_SEARCH_MAX_KP + 32) / 32];
    const int lane = threadIdx.x, f = blockIdx.x;
    const int n = D.n_arr ? min(D.n_arr[f], D.kp_cap) : D.n_fixed;
    const int m = D.m_arr ? min(D.m_arr[f], D.q_cap) : D.m_fixed;
    const uint8_t* TK = D.kp_taken + (size_t)f * D.kp_cap;
    for (int w = lane; w < (n + 31) / 32; w += 64) {
        unsigned bits = 0;
        for (int b = 0; b < 32; ++b) { const int i = 32 * w + b; if (i < n && TK[i]) bits |= 1u << b; }
        taken[w] = bits;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long NONE = ~0ull;
    const size_t q0 = (size_t)f * D.q_cap;
    int nmatches = 0;
    for (int qb = 0; qb < m; qb += 64) {
        const int qj = qb + lane;
        const bool live = qj < m;
        unsigned long long tb = NONE, ts = NONE;
        int ncq = 0, tkq = 0;
        if (live) { tb = D.tent[2 * (q0 + qj)]; ts = D.tent[2 * (q0 + qj) + 1]; ncq = D.cand_n[q0 + qj]; tkq = D.takes[q0 + qj]; }
        int res = -1, bd = 256, bl = -1, sd = 256, sl = -1;     // lane j collects query qb + j
        const int jn = min(64, m - qb);
        for (int j = 0; j < jn; ++j) {
            unsigned long long best = readlane_u64(tb, j), second = readlane_u64(ts, j);
            const int takes_q = __builtin_amdgcn_readlane(tkq, j);
            if (best != NONE) {
                const int bi0 = KEY_KP(best), si0 = second != NONE ? KEY_KP(second) : bi0;
                const bool stale = (((taken[bi0 >> 5] >> (bi0 & 31)) | (taken[si0 >> 5] >> (si0 & 31))) & 1u) != 0;
                if (stale) {   // an earlier query of this call took one of the two: re-scan this query's candidates
                    const size_t qo = q0 + qb + j;
                    const int nc_all = __builtin_amdgcn_readlane(ncq, j);
                    // The stored list is only read HERE.  The tentative pair was reduced over every hit, so a window that holds
                    // more than cand_cap keypoints is exact as long as no re-scan of it is needed; a re-scan of a TRUNCATED list
                    // walks the frame's keypoints again instead (the reference has no limit on a window's size, ORBmatcher.cc:85-117).
                    unsigned long long k1 = NONE, k2 = NONE;
                    if (nc_all <= D.cand_cap) {
                        const unsigned long long* C = D.cand + qo * D.cand_cap;
                        for (int c = lane; c < nc_all; c += 64) {
                            const unsigned long long k = C[c];
                            const int i = KEY_KP(k);
                            if ((taken[i >> 5] >> (i & 31)) & 1u) continue;   // F.mvpMapPoints[idx] with observations
                            if (k < k1) { k2 = k1; k1 = k; } else if (k < k2) k2 = k;
                        }
                    } else {
                        QueryWin W;
                        query_window(D, qo, W);   // (it met the grid: the query has candidates)
                        const uint8_t* KP = D.kp + (size_t)f * D.kp_cap * D.kp_rec;
                        const uint8_t* KD = D.kp_desc + (size_t)f * D.kp_cap * 32;
                        for (int i = lane; i < n; i += 64) {
                            unsigned long long k;
                            if (!window_key(D, W, KP, KD, i, k) || ((taken[i >> 5] >> (i & 31)) & 1u)) continue;
                            if (k < k1) { k2 = k1; k1 = k; } else if (k < k2) k2 = k;
                        }
                    }
                    best = wave_min_u64(k1);
                    second = wave_min_u64(k1 == best ? k2 : k1);
                }
            }
            int r_res = -1, r_bd = 256, r_bl = -1, r_sd = 256, r_sl = -1;
            if (best != NONE) {
                const int bi = KEY_KP(best);
                r_bd = (int)(best >> 32); r_bl = KEY_OCT(best);
                if (second != NONE) { r_sd = (int)(second >> 32); r_sl = KEY_OCT(second); }
                if (r_bd <= D.th_dist) {   // ORBmatcher.cc:120-128
                    const bool reject = D.use_ratio && r_bl == r_sl && (float)r_bd > D.nnratio * (float)r_sd;
                    if (!reject) {
                        r_res = bi;
                        if (takes_q) {
                            if (lane == 0) taken[bi >> 5] |= 1u << (bi & 31);
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                        }
                        ++nmatches;
                    }
                }
            }
            if (lane == j) { res = r_res; bd = r_bd; bl = r_bl; sd = r_sd; sl = r_sl; }
        }
        if (live) {
            const size_t qo = q0 + qj;
            D.match_kp[qo] = res;
            if (D.out4) *reinterpret_cast<int4*>(&D.out4[4 * qo]) = make_int4(bd, bl, sd, sl);
        }
    }
    if (lane == 0) D.nmatches[f] = nmatches;
}

// mode 1 (ORBmatcher::SearchForInitialization): the per-keypoint state is the distance of its current match and the query
// that holds it.  Same speculation: the tentative pair is exact unless one of the two is excluded by the state
// (matched distance <= this query's distance to it); otherwise the query's candidates are re-scanned with the exclusion.
__global__ __launch_bounds__(64) void search_resolve_init_kernel(SearchDev D) {
    extern __shared__ int s_state[];   // md[kp_cap] | m21[kp_cap]
    const int lane = threadIdx.x, f = blockIdx.x;
    const int n = D.n_arr ? min(D.n_arr[f], D.kp_cap) : D.n_fixed;
    const int m = D.m_arr ? min(D.m_arr[f], D.q_cap) : D.m_fixed;
    int* md = s_state;
    int* m21 = s_state + D.kp_cap;
    for (int i = lane; i < n; i += 64) { md[i] = 0x7FFFFFFF; m21[i] = -1; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned long long NONE = ~0ull;
    const size_t q0 = (size_t)f * D.q_cap;
    int nmatches = 0;
    for (int qb = 0; qb < m; qb += 64) {
        const int qj = qb + lane;
        const bool live = qj < m;
        unsigned long long tb = NONE, ts = NONE;
        int ncq = 0;
        if (live) { tb = D.tent[2 * (q0 + qj)]; ts = D.tent[2 * (q0 + qj) + 1]; ncq = D.cand_n[q0 + qj]; }
        int res = -1, bd = 256, bl = -1, sd = 256, sl = -1;
        const int jn = min(64, m - qb);
        for (int j = 0; j < jn; ++j) {
            unsigned long long best = readlane_u64(tb, j), second = readlane_u64(ts, j);
            if (best != NONE) {
                const bool xb = md[KEY_KP(best)] <= (int)(best >> 32);
                const bool xs = second != NONE && md[KEY_KP(second)] <= (int)(second >> 32);
                if (xb || xs) {
                    const size_t qo = q0 + qb + j;
                    const int nc_all = __builtin_amdgcn_readlane(ncq, j);
                    unsigned long long k1 = NONE, k2 = NONE;
                    if (nc_all <= D.cand_cap) {
                        const unsigned long long* C = D.cand + qo * D.cand_cap;
                        for (int c = lane; c < nc_all; c += 64) {
                            const unsigned long long k = C[c];
                            if (md[KEY_KP(k)] <= (int)(k >> 32)) continue;   // ORBmatcher.cc:448
                            if (k < k1) { k2 = k1; k1 = k; } else if (k < k2) k2 = k;
                        }
                    } else {   // a truncated list: walk the frame's keypoints again (see search_resolve_kernel)
                        QueryWin W;
                        query_window(D, qo, W);
                        const uint8_t* KP = D.kp + (size_t)f * D.kp_cap * D.kp_rec;
                        const uint8_t* KD = D.kp_desc + (size_t)f * D.kp_cap * 32;
                        for (int i = lane; i < n; i += 64) {
                            unsigned long long k;
                            if (!window_key(D, W, KP, KD, i, k) || md[i] <= (int)(k >> 32)) continue;
                            if (k < k1) { k2 = k1; k1 = k; } else if (k < k2) k2 = k;
                        }
                    }
                    best = wave_min_u64(k1);
                    second = wave_min_u64(k1 == best ? k2 : k1);
                }
            }
            int r_res = -1, r_bd = 256, r_bl = -1, r_sd = 256, r_sl = -1;
            if (best != NONE) {
                const int bi = KEY_KP(best);
                r_bd = (int)(best >> 32); r_bl = -1;
                float second_f = 2147483648.0f;   // (float)INT_MAX
                if (second != NONE) { r_sd = (int)(second >> 32); r_sl = KEY_OCT(second); second_f = (float)r_sd; }
                if (r_bd <= D.th_dist && (float)r_bd < second_f * D.nnratio) {   // ORBmatcher.cc:462-464
                    const int old = m21[bi];
                    if (old >= 0) {   // the keypoint changes hands: vnMatches12[vnMatches21[bestIdx2]] = -1
                        if (old >= qb) { if (lane == old - qb) res = -1; }
                        else if (lane == 0) D.match_kp[q0 + old] = -1;
                        --nmatches;
                    }
                    r_res = bi;
                    r_bl = bi;   // mode 1 reports the keypoint accepted AT DECISION TIME in the level slot (levels are all 0
                                 // here); unlike match_kp it is not reset by a later take-over -- the rotation histogram bins it
                    if (lane == 0) { m21[bi] = qb + j; md[bi] = r_bd; }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    ++nmatches;
                }
            }
            if (lane == j) { res = r_res; bd = r_bd; bl = r_bl; sd = r_sd; sl = r_sl; }
        }
        if (live) {
            const size_t qo = q0 + qj;
            D.match_kp[qo] = res;
            if (D.out4) *reinterpret_cast<int4*>(&D.out4[4 * qo]) = make_int4(bd, bl, sd, sl);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");   // later chunks may reset entries of this one through global memory
    }
    if (lane == 0) D.nmatches[f] = nmatches;
}

static void search_launch(hipStream_t st, const SearchDev& D, int max_m) {
    if (max_m > 0)
        hipLaunchKernelGGL(search_candidates_kernel, dim3((max_m + 3) / 4, D.nframes), dim3(256), 0, st, D);
    if (D.mode == 1) {
        static bool prepared = false;   // 2 x 4 x 8191 bytes of state sit just under the 64 KB default; ask explicitly
        if (!prepared) { hipFuncSetAttribute(reinterpret_cast<const void*>(search_resolve_init_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024); prepared = true; }
        hipLaunchKernelGGL(search_resolve_init_kernel, dim3(D.nframes), dim3(64), 2 * sizeof(int) * (size_t)D.kp_cap, st, D);
    } else
        hipLaunchKernelGGL(search_resolve_kernel, dim3(D.nframes), dim3(64), 0, st, D);
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
    SLAMIT_USE_DEVICE(device);
    const int n = F->n, m = Q->m, cap = std::min(std::max(n, 1), SLAMIT_SEARCH_MAX_CAND);
    // One pinned staging block and one device slab per host thread, kept between calls (a Tracking thread makes this
    // call every frame: a fresh hipMalloc + nine pageable copies cost more than the search itself).  Layout of both:
    // inputs (keypoint records | descriptors | taken | queries ...) first, then outputs; the candidate lists and
    // tentative pairs live only on the device.
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~(size_t)255; return o; };
    const size_t o_kp = take(12 * (size_t)n), o_kd = take(32 * (size_t)n), o_tk = take((size_t)n);
    const size_t o_uvr = take(12 * (size_t)m), o_l0 = take(4 * (size_t)m), o_l1 = take(4 * (size_t)m), o_qd = take(32 * (size_t)m), o_va = take((size_t)m), o_tq = take((size_t)m);
    const size_t in_bytes = off;
    const size_t o_mk = take(4 * (size_t)m), o_o4 = take(16 * (size_t)m), o_nm = take(4);
    const size_t io_bytes = off;
    const size_t o_cand = take(8 * (size_t)m * cap), o_cn = take(4 * (size_t)m), o_te = take(16 * (size_t)m);
    static thread_local SlamitScratch S;
    {
        const hipError_t es = slamit_scratch_reserve(S, device, io_bytes, off);   // pinned: inputs + outputs; device: + candidate lists
        if (es != hipSuccess) return slamit_fail_hip(es, "slamit_guided_search: scratch");
    }
    uint8_t* hb = S.host;
    uint8_t* d = S.dev;
    for (int i = 0; i < n; ++i) {   // keypoints as {x, y, octave} records
        memcpy(hb + o_kp + 12 * (size_t)i, &F->kp_xy[2 * i], 8);
        memcpy(hb + o_kp + 12 * (size_t)i + 8, &F->kp_octave[i], 4);
    }
    memcpy(hb + o_kd, F->desc, 32 * (size_t)n); memcpy(hb + o_tk, F->kp_taken, (size_t)n);
    memcpy(hb + o_uvr, Q->uvr, 12 * (size_t)m); memcpy(hb + o_l0, Q->level_min, 4 * (size_t)m); memcpy(hb + o_l1, Q->level_max, 4 * (size_t)m);
    memcpy(hb + o_qd, Q->desc, 32 * (size_t)m); memcpy(hb + o_va, Q->valid, (size_t)m);
    if (Q->takes) memcpy(hb + o_tq, Q->takes, (size_t)m); else memset(hb + o_tq, 1, (size_t)m);
    hipError_t e = hipMemcpyAsync(d, hb, in_bytes, hipMemcpyHostToDevice, S.st);
    if (e == hipSuccess) {
        SearchDev D;
        D.nframes = 1; D.kp_cap = std::max(n, 1); D.q_cap = m; D.cand_cap = cap;
        D.n_arr = nullptr; D.n_fixed = n; D.m_arr = nullptr; D.m_fixed = m;
        D.kp = d + o_kp; D.kp_rec = 12; D.kp_oct_off = 8; D.kp_desc = d + o_kd; D.kp_taken = d + o_tk;
        D.min_x = F->min_x; D.min_y = F->min_y; D.inv_w = F->inv_w; D.inv_h = F->inv_h;
        D.uvr = (const float*)(d + o_uvr); D.lmin = (const int*)(d + o_l0); D.lmax = (const int*)(d + o_l1); D.qdesc = d + o_qd; D.valid = d + o_va; D.takes = d + o_tq;
        D.cand = (unsigned long long*)(d + o_cand); D.cand_n = (int*)(d + o_cn); D.tent = (unsigned long long*)(d + o_te);
        D.th_dist = rule->th_dist; D.use_ratio = rule->use_ratio; D.nnratio = rule->nnratio;
        D.chi2_gate = rule->mode == 1 ? 0.f : rule->chi2_gate; memcpy(D.inv_sigma2, rule->inv_level_sigma2, sizeof(D.inv_sigma2));
        D.mode = rule->mode;
        D.match_kp = (int*)(d + o_mk); D.out4 = (int*)(d + o_o4); D.nmatches = (int*)(d + o_nm);
        search_launch(S.st, D, m);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hb + in_bytes, d + in_bytes, io_bytes - in_bytes, hipMemcpyDeviceToHost, S.st);
    if (e == hipSuccess) e = hipStreamSynchronize(S.st);
    if (e != hipSuccess) return slamit_fail_hip(e, "slamit_guided_search");
    int nm = 0;
    memcpy(&nm, hb + o_nm, 4);
    memcpy(match_kp, hb + o_mk, 4 * (size_t)m);
    *nmatches = nm;
    const int* o4 = reinterpret_cast<const int*>(hb + o_o4);
    for (int q = 0; q < m; ++q) {
        if (best_dist) best_dist[q] = o4[4 * (size_t)q];
        if (best_level) best_level[q] = o4[4 * (size_t)q + 1];
        if (second_dist) second_dist[q] = o4[4 * (size_t)q + 2];
        if (second_level) second_level[q] = o4[4 * (size_t)q + 3];
    }
    return SLAMIT_OK;
}

extern "C" size_t slamit_guided_search_workspace(int nframes, int q_cap) {
    if (nframes < 0 || q_cap < 0) return 0;
    return (size_t)nframes * q_cap * (8 * (size_t)SLAMIT_SEARCH_BATCH_CAND + 16 + 4) + 256;
}

extern "C" int slamit_guided_search_batch_dev(int device, const slamit_search_batch* B, const slamit_search_rule* rule,
                                              int32_t* d_match_kp, int32_t* d_nmatches, int32_t* d_out4, void* d_workspace,
                                              size_t workspace_bytes, void* stream) {
    if (!B || !rule || !d_match_kp || !d_nmatches || B->nframes < 0 || B->kp_cap < 0 || B->q_cap < 0)
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_guided_search_batch_dev: bad argument");
    if (B->nframes == 0) return SLAMIT_OK;
    if (!B->d_n || !B->d_kps_un || !B->d_desc || !B->d_kp_taken || !B->d_m || !B->d_uvr || !B->d_level_min || !B->d_level_max ||
        !B->d_qdesc || !B->d_valid || !B->d_takes || !d_workspace)
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_guided_search_batch_dev: null array");
    if (B->kp_cap > SLAMIT_SEARCH_MAX_KP) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_guided_search_batch_dev: kp_cap > SLAMIT_SEARCH_MAX_KP");
    if (workspace_bytes < slamit_guided_search_workspace(B->nframes, B->q_cap))
        return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_guided_search_batch_dev: workspace smaller than slamit_guided_search_workspace()");
    SLAMIT_USE_DEVICE(device);
    SearchDev D;
    D.nframes = B->nframes; D.kp_cap = B->kp_cap; D.q_cap = B->q_cap; D.cand_cap = SLAMIT_SEARCH_BATCH_CAND;
    D.n_arr = B->d_n; D.n_fixed = 0; D.m_arr = B->d_m; D.m_fixed = 0;
    D.kp = reinterpret_cast<const uint8_t*>(B->d_kps_un); D.kp_rec = (int)sizeof(slamit_kp); D.kp_oct_off = 20;
    D.kp_desc = B->d_desc; D.kp_taken = B->d_kp_taken;
    D.min_x = B->min_x; D.min_y = B->min_y; D.inv_w = B->inv_w; D.inv_h = B->inv_h;
    D.uvr = B->d_uvr; D.lmin = B->d_level_min; D.lmax = B->d_level_max; D.qdesc = B->d_qdesc; D.valid = B->d_valid; D.takes = B->d_takes;
    const size_t nq = (size_t)B->nframes * B->q_cap;
    D.cand = reinterpret_cast<unsigned long long*>(d_workspace);
    D.tent = D.cand + nq * SLAMIT_SEARCH_BATCH_CAND;
    D.cand_n = reinterpret_cast<int*>(D.tent + 2 * nq);
    D.th_dist = rule->th_dist; D.use_ratio = rule->use_ratio; D.nnratio = rule->nnratio;
    D.chi2_gate = rule->mode == 1 ? 0.f : rule->chi2_gate; memcpy(D.inv_sigma2, rule->inv_level_sigma2, sizeof(D.inv_sigma2));
    D.mode = rule->mode;
    D.match_kp = d_match_kp; D.out4 = d_out4; D.nmatches = d_nmatches;
    search_launch((hipStream_t)stream, D, B->q_cap);
    HIP_TRY(hipGetLastError());
    return SLAMIT_OK;
}
