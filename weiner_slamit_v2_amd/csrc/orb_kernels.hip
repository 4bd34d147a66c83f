// orb_kernels.hip — hand-written gfx950 kernels of the ORB extractor hot path.
//
// Replaces, per batch of frames resident in HBM (reference = ORB_SLAM2/src/ORBextractor.cc):
//   K1 resize_level_kernel      ComputePyramid :1138-1168 (cv::resize 8UC1 INTER_LINEAR)
//   K2 fast_cells_kernel        ComputeKeyPointsOctTree :805-849 (cv::FAST per 30x30 cell,
//                               iniThFAST then minThFAST fallback, NMS inside the cell window)
//   K4 octree_kernel            DistributeOctTree :552-776
//   K5 ic_angle_kernel          IC_Angle :82-109
//   K6 blur_level_kernel        GaussianBlur 7x7 sigma 2, REFLECT_101 :1116-1117
//   K7 describe_kernel          computeOrbDescriptor :113-152 + keypoint scaling/concat :1126-1134
//   pad_level_kernel            copyMakeBorder REFLECT_101 (only for the mvImagePyramid getter)
//
// All integer/byte arithmetic is exact; the only floating point is slamit_math.h, compiled with
// -ffp-contract=off.  Wavefront = 64 everywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orb_types.h"
#include "slamit_math.h"
#include "../../include/slamit.h"
#include "../../include/slamit_orb_pattern.h"

#define WAVE 64

// --------------------------------------------------------------------------------------------
// K1: bilinear 8-bit resize with OpenCV's fixed-point coefficients.  xofs/ialpha (per dst column)
// and yofs/ibeta (per dst row) are built once per handle on the host with the same float/double
// steps OpenCV uses; the kernel does the two integer passes.  One thread = 4 adjacent dst pixels
// (one aligned dword store; dst rows start on 64-byte boundaries).
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_level_kernel(
    const uint8_t* __restrict__ src, int sw, int sh, size_t sstride, size_t sframe,
    uint8_t* __restrict__ dst, int dw, int dh, size_t dstride, size_t dframe,
    const int* __restrict__ xofs, const short* __restrict__ ialpha,
    const int* __restrict__ yofs, const short* __restrict__ ibeta) {
    const int x4 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int f = blockIdx.z;
    if (x4 >= dw || y >= dh) return;
    const uint8_t* S = src + (size_t)f * sframe;
    int sy = yofs[y];
    int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
    const uint8_t* S0 = S + (size_t)sy0 * sstride;
    const uint8_t* S1 = S + (size_t)sy1 * sstride;
    int b0 = ibeta[2 * y], b1 = ibeta[2 * y + 1];
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int x = x4 + i;
        if (x < dw) {
            int sx = xofs[x];
            int sx1 = min(sx + 1, sw - 1);
            int a0 = ialpha[2 * x], a1 = ialpha[2 * x + 1];
            int r0 = S0[sx] * a0 + S0[sx1] * a1;
            int r1 = S1[sx] * a0 + S1[sx1] * a1;
            int v = ((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2) & 0xFF;
            out |= (uint32_t)v << (8 * i);
        }
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)f * dframe + (size_t)y * dstride + x4) = out;
}

// --------------------------------------------------------------------------------------------
// K2: FAST-9/16 per cell.  One 256-thread workgroup = one cell of one level of one frame.
//
// The cell's window (<= 65x65 bytes) is staged in LDS once; every pixel of the scan area gets its
// INTRINSIC score S = max over the sixteen 9-arcs of min |v - ring| (same sign) - 1, which is what
// cornerScore<16> returns for any threshold t <= S, and "corner at t" <=> S >= t.  Sliding
// 9-window max/min over the ring are built from 3-input max/min (v_max3/v_min3).  NMS compares
// against the 8 neighbours' scores inside the cell's scan area only (cv::FAST zero-fills outside),
// then the workgroup decides iniThFAST vs minThFAST and appends its keypoints to the
// (frame, level) candidate list.  A candidate is one u64: (score << 32) | order, where
// order = (cell << 12) | (y_local << 6) | x_local is the position in the reference's
// vToDistributeKeys order (cell row, cell col, y, x); the octree only needs that order to break
// response ties, so the list itself may be unordered.
// --------------------------------------------------------------------------------------------
#define TILE_PITCH 72
#define SC_PITCH 64

__device__ __forceinline__ int imax3(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int imin3(int a, int b, int c) { return min(min(a, b), c); }

__device__ __forceinline__ int fast_score(const uint8_t* t) {
    // ring in the order of cv::FAST's 16-pattern: (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)
    // (0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
    int r[16];
    r[0] = t[3 * TILE_PITCH + 0];   r[1] = t[3 * TILE_PITCH + 1];   r[2] = t[2 * TILE_PITCH + 2];
    r[3] = t[1 * TILE_PITCH + 3];   r[4] = t[3];                    r[5] = t[-1 * TILE_PITCH + 3];
    r[6] = t[-2 * TILE_PITCH + 2];  r[7] = t[-3 * TILE_PITCH + 1];  r[8] = t[-3 * TILE_PITCH + 0];
    r[9] = t[-3 * TILE_PITCH - 1];  r[10] = t[-2 * TILE_PITCH - 2]; r[11] = t[-1 * TILE_PITCH - 3];
    r[12] = t[-3];                  r[13] = t[1 * TILE_PITCH - 3];  r[14] = t[2 * TILE_PITCH - 2];
    r[15] = t[3 * TILE_PITCH - 1];
    const int v = t[0];
    int hi3[16], lo3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        hi3[k] = imax3(r[k], r[(k + 1) & 15], r[(k + 2) & 15]);
        lo3[k] = imin3(r[k], r[(k + 1) & 15], r[(k + 2) & 15]);
    }
    int dm = 255, bm = 0;  // min over arcs of the arc's max; max over arcs of the arc's min
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        dm = min(dm, imax3(hi3[k], hi3[(k + 3) & 15], hi3[(k + 6) & 15]));
        bm = max(bm, imin3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]));
    }
    return max(v - dm, bm - v) - 1;
}

__global__ __launch_bounds__(256) void fast_cells_kernel(
    const OrbLevel* __restrict__ levels, int nlevels,
    const uint8_t* __restrict__ img0, size_t img0_stride, size_t img0_frame,
    const uint8_t* __restrict__ pyr,
    unsigned long long* __restrict__ cand, size_t cand_frame_stride,
    int* __restrict__ cand_count, int iniTh, int minTh) {
    __shared__ uint8_t tile[ORB_TILE_MAX * TILE_PITCH];
    __shared__ __attribute__((aligned(16))) uint8_t sc[(ORB_CELL_MAX + 2) * SC_PITCH];
    __shared__ int s_cnt_ini, s_cnt_min, s_base;

    const int tid = threadIdx.x;
    const int frame = blockIdx.y;
    const int cell = blockIdx.x;
    int level = 0;
    while (level + 1 < nlevels && cell >= levels[level + 1].cell_base) ++level;
    const OrbLevel& L = levels[level];
    const int c = cell - L.cell_base;
    const int ci = c / L.nCols, cj = c - ci * L.nCols;
    const int iniX = ORB_MIN_BORDER + cj * L.wCell, iniY = ORB_MIN_BORDER + ci * L.hCell;
    if (iniY >= L.maxBorderY - 3 || iniX >= L.maxBorderX - 6) return;  // ORBextractor.cc:810,819
    const int cw = min(L.wCell + 6, L.maxBorderX - iniX);
    const int ch = min(L.hCell + 6, L.maxBorderY - iniY);
    const int sw = cw - 6, sh = ch - 6;  // scan area = FAST's [3, n-3)
    if (sw <= 0 || sh <= 0) return;

    const uint8_t* src;
    size_t stride;
    if (level == 0) {
        src = img0 + (size_t)frame * img0_frame;
        stride = img0_stride;
    } else {
        src = pyr + L.plane_off + (size_t)frame * L.plane_bytes;
        stride = (size_t)L.stride;
    }
    src += (size_t)iniY * stride + iniX;

    // stage the window; rows are <= 65 contiguous bytes, neighbouring cells share L2 lines
    const int npx = cw * ch;
    const unsigned inv_cw = (1u << 20) / (unsigned)cw + 1;  // exact p / cw for p*cw < 2^20
    for (int p = tid; p < npx; p += 256) {
        int r = (int)(((unsigned)p * inv_cw) >> 20);
        int cc = p - r * cw;
        tile[r * TILE_PITCH + cc] = src[(size_t)r * stride + cc];
    }
    for (int i = tid; i < (ORB_CELL_MAX + 2) * SC_PITCH / 4; i += 256) reinterpret_cast<uint32_t*>(sc)[i] = 0;
    if (tid == 0) { s_cnt_ini = 0; s_cnt_min = 0; }
    __syncthreads();

    const int floorTh = max(min(iniTh, minTh), 1);
    const int nscan = sw * sh;
    const unsigned inv_sw = (1u << 20) / (unsigned)sw + 1;
    for (int p = tid; p < nscan; p += 256) {
        int y = (int)(((unsigned)p * inv_sw) >> 20);
        int x = p - y * sw;
        int S = fast_score(&tile[(y + 3) * TILE_PITCH + x + 3]);
        sc[(y + 1) * SC_PITCH + x + 1] = (uint8_t)(S >= floorTh ? S : 0);
    }
    __syncthreads();

    // NMS (strict >, 8 neighbours, zeros outside the scan area); remember per-thread results
    uint32_t m_ini = 0, m_min = 0;
    int it = 0;
    for (int p = tid; p < nscan; p += 256, ++it) {
        int y = (int)(((unsigned)p * inv_sw) >> 20);
        int x = p - y * sw;
        const uint8_t* s = &sc[(y + 1) * SC_PITCH + x + 1];
        int v = s[0];
        if (v) {
            int nb = imax3(s[-SC_PITCH - 1], s[-SC_PITCH], s[-SC_PITCH + 1]);
            nb = imax3(nb, s[-1], s[1]);
            nb = max(nb, imax3(s[SC_PITCH - 1], s[SC_PITCH], s[SC_PITCH + 1]));
            if (v > nb) {
                if (v >= iniTh) m_ini |= 1u << it;
                if (v >= minTh) m_min |= 1u << it;
            }
        }
    }
    int n_ini = __popc(m_ini), n_min = __popc(m_min);
    int off_ini = 0, off_min = 0;
    if (n_ini) off_ini = atomicAdd(&s_cnt_ini, n_ini);
    if (n_min) off_min = atomicAdd(&s_cnt_min, n_min);
    __syncthreads();
    const bool use_ini = s_cnt_ini > 0;  // ORBextractor.cc:829-833: retry at minThFAST if empty
    const int total = use_ini ? s_cnt_ini : s_cnt_min;
    if (total == 0) return;
    if (tid == 0) s_base = atomicAdd(&cand_count[frame * nlevels + level], total);
    __syncthreads();
    uint32_t m = use_ini ? m_ini : m_min;
    int o = s_base + (use_ini ? off_ini : off_min);
    unsigned long long* out = cand + L.cand_off + (size_t)frame * cand_frame_stride;
    while (m) {
        int b = __ffs(m) - 1;
        m &= m - 1;
        int p = tid + 256 * b;
        int y = (int)(((unsigned)p * inv_sw) >> 20);
        int x = p - y * sw;
        unsigned S = sc[(y + 1) * SC_PITCH + x + 1];
        unsigned order = ((unsigned)c << 12) | ((unsigned)(y + 3) << 6) | (unsigned)(x + 3);
        if (o < L.cand_cap) out[o] = ((unsigned long long)S << 32) | order;
        ++o;
    }
}

// --------------------------------------------------------------------------------------------
// K4: DistributeOctTree.  One workgroup per (frame, level).  Nodes are kept in LDS in LIST ORDER
// (front of the reference's std::list first): children are pushed to the front, so after a pass
// the list is reverse(creation order of the new children) ++ (surviving old nodes in old order).
// Key-level work (quadrant counting, re-labelling) is parallel over the candidates; node-level
// bookkeeping is done with block-wide prefix sums.  The final pick per node is
// max (response, then earliest position in vToDistributeKeys) via a packed u64 LDS atomicMax.
// Tie-break of the "largest first" phase: (size desc, creation seq desc) == (size desc, list
// position asc); see DESIGN.md "octree determinism".
// --------------------------------------------------------------------------------------------
struct Box16 { short x0, y0, x1, y1; };

__device__ __forceinline__ int box_quadrant(const Box16& b, int x, int y, int* mx, int* my) {
    int midx = b.x0 + ((b.x1 - b.x0 + 1) >> 1);  // UL.x + ceil((UR.x-UL.x)/2)
    int midy = b.y0 + ((b.y1 - b.y0 + 1) >> 1);
    *mx = midx; *my = midy;
    return (x < midx ? 0 : 1) + (y < midy ? 0 : 2);  // 0:n1 1:n2 2:n3 3:n4
}
__device__ __forceinline__ Box16 child_box(const Box16& b, int q) {
    int midx = b.x0 + ((b.x1 - b.x0 + 1) >> 1);
    int midy = b.y0 + ((b.y1 - b.y0 + 1) >> 1);
    Box16 c;
    c.x0 = (q & 1) ? midx : b.x0; c.x1 = (q & 1) ? b.x1 : midx;
    c.y0 = (q & 2) ? midy : b.y0; c.y1 = (q & 2) ? b.y1 : midy;
    return c;
}

// exclusive scan of a[0..len) in place (LDS), returns the total; all 256 threads call it
__device__ int block_exclusive_scan(int* a, int len, int* wave_tmp /*[4+1]*/) {
    const int tid = threadIdx.x;
    const int C = (len + 255) / 256;
    const int lo = min(tid * C, len), hi = min(lo + C, len);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += a[i];
    // inclusive scan of `sum` across the wave
    int incl = sum;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        int t = __shfl_up(incl, d, WAVE);
        if ((tid & 63) >= d) incl += t;
    }
    if ((tid & 63) == 63) wave_tmp[tid >> 6] = incl;
    __syncthreads();
    int wbase = 0;
    for (int w = 0; w < (tid >> 6); ++w) wbase += wave_tmp[w];
    int total = wave_tmp[0] + wave_tmp[1] + wave_tmp[2] + wave_tmp[3];
    int run = wbase + incl - sum;
    for (int i = lo; i < hi; ++i) {
        int v = a[i];
        a[i] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

__global__ __launch_bounds__(256) void octree_kernel(
    const OrbLevel* __restrict__ levels, int nlevels,
    const unsigned long long* __restrict__ cand, size_t cand_frame_stride,
    const int* __restrict__ cand_count,
    uint32_t* __restrict__ ws_xy, uint16_t* __restrict__ ws_node,
    OrbLevelKp* __restrict__ lkp, size_t kp_frame_stride, int* __restrict__ kp_count,
    int node_cap, int level_override /* -1: blockIdx.x */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int level = level_override >= 0 ? level_override : blockIdx.x;
    const int frame = blockIdx.y;
    const OrbLevel& L = levels[level];
    const int cap = node_cap;
    // LDS carve (all 8-byte aligned)
    unsigned long long* best = reinterpret_cast<unsigned long long*>(smem);              // cap
    Box16* box[2] = {reinterpret_cast<Box16*>(best + cap), reinterpret_cast<Box16*>(best + cap) + cap};
    int* cnt[2] = {reinterpret_cast<int*>(box[1] + cap), reinterpret_cast<int*>(box[1] + cap) + cap};
    int* childcnt = cnt[1] + cap;        // 4*cap
    int* scanbuf = childcnt + 4 * cap;   // 5*cap
    int* rankv = scanbuf + 5 * cap;      // cap   (rank of an expandable node in the sorted order)
    int* sorted = rankv + cap;           // cap   (node index at each rank)
    __shared__ int wave_tmp[8];
    __shared__ int s_n, s_expand, s_k, s_flag;

    const int kidx = frame * nlevels + level;
    const int n_keys = min(cand_count[kidx], L.cand_cap);
    const int N = L.quota;
    const unsigned long long* K = cand + L.cand_off + (size_t)frame * cand_frame_stride;
    uint32_t* XY = ws_xy + L.cand_off + (size_t)frame * cand_frame_stride;
    uint16_t* ND = ws_node + L.cand_off + (size_t)frame * cand_frame_stride;
    OrbLevelKp* OUT = lkp + L.kp_off + (size_t)frame * kp_frame_stride;
    if (n_keys == 0) {
        if (tid == 0) kp_count[kidx] = 0;
        return;
    }

    // ---- roots (ORBextractor.cc:556-598) ----
    const int nIni = L.nIni;
    if (tid < nIni) {
        Box16 b; b.x0 = (short)L.rootUL[tid]; b.y0 = 0; b.x1 = (short)L.rootUR[tid]; b.y1 = (short)L.boxH;
        box[0][tid] = b;
        cnt[0][tid] = 0;
    }
    __syncthreads();
    for (int k = tid; k < n_keys; k += 256) {
        unsigned order = (unsigned)K[k];
        int lx = order & 63, ly = (order >> 6) & 63, cell = order >> 12;
        int ci = cell / L.nCols, cj = cell - ci * L.nCols;
        int x = cj * L.wCell + lx, y = ci * L.hCell + ly;
        XY[k] = (uint32_t)x | ((uint32_t)y << 16);
        int r = (int)((float)x / L.hX);  // vpIniNodes[kp.pt.x/hX]
        r = min(r, nIni - 1);
        ND[k] = (uint16_t)r;
        atomicAdd(&cnt[0][r], 1);
    }
    __syncthreads();
    if (tid == 0) {  // erase empty roots, keep order (<= 8 roots)
        int m = 0, dropped = 0;
        for (int i = 0; i < nIni; ++i) {
            if (cnt[0][i] > 0) { scanbuf[i] = m; box[0][m] = box[0][i]; cnt[0][m] = cnt[0][i]; ++m; }
            else { scanbuf[i] = -1; dropped = 1; }
        }
        s_n = m; s_flag = dropped;
    }
    __syncthreads();
    if (s_flag) {
        for (int k = tid; k < n_keys; k += 256) ND[k] = (uint16_t)scanbuf[ND[k]];
        __syncthreads();
    }

    int cur = 0;
    int n = s_n;
    bool finish = false, careful = false;
    while (!finish) {
        const int prevSize = n;
        // children populations of every expandable node
        for (int i = tid; i < 4 * n; i += 256) childcnt[i] = 0;
        __syncthreads();
        for (int k = tid; k < n_keys; k += 256) {
            int nd = ND[k];
            if (cnt[cur][nd] > 1) {
                uint32_t xy = XY[k];
                int mx, my;
                int q = box_quadrant(box[cur][nd], xy & 0xFFFF, xy >> 16, &mx, &my);
                atomicAdd(&childcnt[nd * 4 + q], 1);
            }
        }
        __syncthreads();

        int kproc;  // number of parents split in this pass, in processing order
        if (!careful) {
            // every expandable node is split, in list order
            kproc = -1;
        } else {
            // "largest first": rank expandable nodes by (population desc, list position asc)
            for (int i = tid; i < n; i += 256) {
                int ci_ = cnt[cur][i];
                int r = -1;
                if (ci_ > 1) {
                    r = 0;
                    for (int j = 0; j < n; ++j) {
                        int cj_ = cnt[cur][j];
                        r += (cj_ > 1) && (cj_ > ci_ || (cj_ == ci_ && j < i));
                    }
                    sorted[r] = i;
                }
                rankv[i] = r;
            }
            if (tid == 0) s_expand = 0;
            __syncthreads();
            // m = number of expandable nodes
            int local = 0;
            for (int i = tid; i < n; i += 256) local += cnt[cur][i] > 1;
            if (local) atomicAdd(&s_expand, local);
            __syncthreads();
            const int m = s_expand;
            // gains in sorted order, prefix, first position where the list reaches N
            for (int s = tid; s < m; s += 256) {
                int i = sorted[s];
                const int* cc = &childcnt[i * 4];
                scanbuf[s] = (cc[0] > 0) + (cc[1] > 0) + (cc[2] > 0) + (cc[3] > 0) - 1;
            }
            if (tid == 0) s_k = m;
            __syncthreads();
            block_exclusive_scan(scanbuf, m, wave_tmp);  // scanbuf[s] = gain of ranks < s
            for (int s = tid; s < m; s += 256) {
                int i = sorted[s];
                const int* cc = &childcnt[i * 4];
                int g = (cc[0] > 0) + (cc[1] > 0) + (cc[2] > 0) + (cc[3] > 0) - 1;
                if (n + scanbuf[s] + g >= N) atomicMin(&s_k, s + 1);
            }
            __syncthreads();
            kproc = s_k;
        }

        // flags of the new list: children of split parents in reverse processing order (n4..n1),
        // then the nodes that stay, in their old order
        int nchildslots;
        if (!careful) {
            nchildslots = 4 * n;
            for (int e = tid; e < 4 * n; e += 256) {
                int i = n - 1 - (e >> 2), q = 3 - (e & 3);
                scanbuf[e] = (cnt[cur][i] > 1) && (childcnt[i * 4 + q] > 0);
            }
            for (int i = tid; i < n; i += 256) scanbuf[4 * n + i] = cnt[cur][i] <= 1;
        } else {
            nchildslots = 4 * kproc;
            for (int e = tid; e < nchildslots; e += 256) {
                int s = kproc - 1 - (e >> 2), q = 3 - (e & 3);
                scanbuf[e] = childcnt[sorted[s] * 4 + q] > 0;
            }
            for (int i = tid; i < n; i += 256) {
                int r = rankv[i];
                scanbuf[nchildslots + i] = !(r >= 0 && r < kproc);
            }
        }
        __syncthreads();
        const int n_new = block_exclusive_scan(scanbuf, nchildslots + n, wave_tmp);
        // (n_new <= cap by construction: a full pass only runs while n + 3*expandable <= N, the
        //  largest-first pass stops within 3 of N; clamp anyway for memory safety)
        const int nxt = cur ^ 1;
        if (tid == 0) s_expand = 0;
        __syncthreads();
        int local_expand = 0;
        for (int e = tid; e < nchildslots; e += 256) {
            int i, q;
            if (!careful) { i = n - 1 - (e >> 2); q = 3 - (e & 3); }
            else { i = sorted[kproc - 1 - (e >> 2)]; q = 3 - (e & 3); }
            bool split = !careful ? (cnt[cur][i] > 1) : true;
            int cc = childcnt[i * 4 + q];
            if (split && cc > 0) {
                int pos = scanbuf[e];
                if (pos < cap) {
                    box[nxt][pos] = child_box(box[cur][i], q);
                    cnt[nxt][pos] = cc;
                }
                local_expand += cc > 1;
            }
        }
        for (int i = tid; i < n; i += 256) {
            bool stays = !careful ? (cnt[cur][i] <= 1) : !(rankv[i] >= 0 && rankv[i] < kproc);
            if (stays) {
                int pos = scanbuf[nchildslots + i];
                if (pos < cap) { box[nxt][pos] = box[cur][i]; cnt[nxt][pos] = cnt[cur][i]; }
            }
        }
        if (local_expand) atomicAdd(&s_expand, local_expand);
        // re-label the keys
        for (int k = tid; k < n_keys; k += 256) {
            int nd = ND[k];
            bool split = !careful ? (cnt[cur][nd] > 1) : (rankv[nd] >= 0 && rankv[nd] < kproc);
            int pos;
            if (split) {
                uint32_t xy = XY[k];
                int mx, my;
                int q = box_quadrant(box[cur][nd], xy & 0xFFFF, xy >> 16, &mx, &my);
                int e = !careful ? ((n - 1 - nd) * 4 + (3 - q)) : ((kproc - 1 - rankv[nd]) * 4 + (3 - q));
                pos = scanbuf[e];
            } else {
                pos = scanbuf[nchildslots + nd];
            }
            ND[k] = (uint16_t)min(pos, cap - 1);
        }
        __syncthreads();
        const int nToExpand = s_expand;
        cur = nxt;
        n = min(n_new, cap);
        __syncthreads();
        if (n >= N || n == prevSize) finish = true;            // ORBextractor.cc:682, 747
        else if (!careful && n + nToExpand * 3 > N) careful = true;  // :686
    }

    // ---- best key per node (ORBextractor.cc:755-773): max response, first in input order ----
    for (int i = tid; i < n; i += 256) best[i] = 0ull;
    __syncthreads();
    for (int k = tid; k < n_keys; k += 256) {
        unsigned long long c = K[k];
        unsigned long long packed = (c & 0xFFFFFFFF00000000ull) | (0xFFFFFFFFu - (unsigned)c);
        atomicMax(&best[ND[k]], packed);
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        unsigned long long b = best[i];
        unsigned order = 0xFFFFFFFFu - (unsigned)b;
        int lx = order & 63, ly = (order >> 6) & 63, cell = order >> 12;
        int ci = cell / L.nCols, cj = cell - ci * L.nCols;
        OrbLevelKp kp;
        kp.x = (int16_t)(cj * L.wCell + lx + ORB_MIN_BORDER);  // ORBextractor.cc:863-864
        kp.y = (int16_t)(ci * L.hCell + ly + ORB_MIN_BORDER);
        kp.response = (float)(unsigned)(b >> 32);
        kp.angle = 0.f;
        if (i < L.kp_cap) OUT[i] = kp;
    }
    if (tid == 0) kp_count[kidx] = min(n, L.kp_cap);
}

// --------------------------------------------------------------------------------------------
// K5: IC_Angle — intensity centroid over the radius-15 disc on the un-blurred level.
// One wavefront per keypoint; integer moments reduced with cross-lane shuffles.
// --------------------------------------------------------------------------------------------
__constant__ int c_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};

__global__ __launch_bounds__(256) void ic_angle_kernel(
    const OrbLevel* __restrict__ levels, int nlevels,
    const uint8_t* __restrict__ img0, size_t img0_stride, size_t img0_frame,
    const uint8_t* __restrict__ pyr,
    OrbLevelKp* __restrict__ lkp, size_t kp_frame_stride, const int* __restrict__ kp_count) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int level = blockIdx.y, frame = blockIdx.z;
    const OrbLevel& L = levels[level];
    if (i >= kp_count[frame * nlevels + level]) return;
    OrbLevelKp* kp = lkp + L.kp_off + (size_t)frame * kp_frame_stride + i;
    const uint8_t* src;
    size_t stride;
    if (level == 0) { src = img0 + (size_t)frame * img0_frame; stride = img0_stride; }
    else { src = pyr + L.plane_off + (size_t)frame * L.plane_bytes; stride = (size_t)L.stride; }
    const uint8_t* center = src + (size_t)kp->y * stride + kp->x;
    int m10 = 0, m01 = 0;
    for (int idx = lane; idx < 31 * 31; idx += WAVE) {
        int v = idx / 31 - 15, u = idx % 31 - 15;
        int av = v < 0 ? -v : v, au = u < 0 ? -u : u;
        if (au <= c_umax[av]) {
            int val = center[(ptrdiff_t)v * (ptrdiff_t)stride + u];
            m10 += u * val;
            m01 += v * val;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        m10 += __shfl_xor(m10, d, WAVE);
        m01 += __shfl_xor(m01, d, WAVE);
    }
    if (lane == 0) kp->angle = slamit_fast_atan2((float)m01, (float)m10);
}

// --------------------------------------------------------------------------------------------
// K6: 7x7 Gaussian blur, fixed point taps [18,34,49,55,49,34,18] (sum 257), REFLECT_101 on the
// un-padded level.  Workgroup = 64x16 output tile; rows pass kept as u16 in LDS (<= 255*257).
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

__global__ __launch_bounds__(256) void blur_level_kernel(
    const uint8_t* __restrict__ src, int w, int h, size_t sstride, size_t sframe,
    uint8_t* __restrict__ dst, size_t dstride, size_t dframe) {
    __shared__ uint8_t in[22][72];
    __shared__ uint16_t rows[22][64];
    const int tid = threadIdx.x;
    const int bx = blockIdx.x * 64, by = blockIdx.y * 16, f = blockIdx.z;
    const uint8_t* S = src + (size_t)f * sframe;
    for (int i = tid; i < 22 * 70; i += 256) {
        int r = i / 70, c = i - r * 70;
        int gy = reflect101(by + r - 3, h), gx = reflect101(bx + c - 3, w);
        in[r][c] = S[(size_t)gy * sstride + gx];
    }
    __syncthreads();
    for (int i = tid; i < 22 * 64; i += 256) {
        int r = i >> 6, c = i & 63;
        const uint8_t* p = &in[r][c];
        int acc = 18 * (p[0] + p[6]) + 34 * (p[1] + p[5]) + 49 * (p[2] + p[4]) + 55 * p[3];
        rows[r][c] = (uint16_t)acc;
    }
    __syncthreads();
    const int x4 = (tid & 15) * 4, y = tid >> 4;
    if (bx + x4 >= w || by + y >= h) return;
    uint32_t out = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int c = x4 + i;
        int acc = 18 * (rows[y][c] + rows[y + 6][c]) + 34 * (rows[y + 1][c] + rows[y + 5][c]) +
                  49 * (rows[y + 2][c] + rows[y + 4][c]) + 55 * rows[y + 3][c];
        int v = (acc + (1 << 15)) >> 16;
        v = min(v, 255);
        out |= (uint32_t)v << (8 * i);
    }
    *reinterpret_cast<uint32_t*>(dst + (size_t)f * dframe + (size_t)(by + y) * dstride + bx + x4) = out;
}

// --------------------------------------------------------------------------------------------
// K7: rotated BRIEF (256 tests) + output assembly.  One wavefront per keypoint: lane l evaluates
// tests 4l..4l+3 (a nibble), lane pairs combine to one descriptor byte.  The wave also writes the
// final cv::KeyPoint-shaped record at its level-major position and lane 0 of keypoint 0 of
// level 0 writes the frame total.
// --------------------------------------------------------------------------------------------
__constant__ signed char c_pattern[SLAMIT_ORB_PATTERN_INTS];

__global__ __launch_bounds__(256) void describe_kernel(
    const OrbLevel* __restrict__ levels, int nlevels,
    const uint8_t* __restrict__ blur,
    const OrbLevelKp* __restrict__ lkp, size_t kp_frame_stride, const int* __restrict__ kp_count,
    slamit_kp* __restrict__ out_kps, uint8_t* __restrict__ out_desc, int out_cap,
    int* __restrict__ out_n) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int level = blockIdx.y, frame = blockIdx.z;
    const OrbLevel& L = levels[level];
    const int* counts = kp_count + frame * nlevels;
    int offset = 0, total = 0;
    for (int l = 0; l < nlevels; ++l) {
        int c = counts[l];
        if (l < level) offset += c;
        total += c;
    }
    if (level == 0 && i == 0 && lane == 0) out_n[frame] = min(total, out_cap);
    if (i >= counts[level]) return;
    const OrbLevelKp kp = lkp[L.kp_off + (size_t)frame * kp_frame_stride + i];
    const int o = offset + i;
    if (o >= out_cap) return;

    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float a, b;
    slamit_sincosf(kp.angle * factorPI, &b, &a);  // a = cos, b = sin   (ORBextractor.cc:117-118)
    const uint8_t* img = blur + L.blur_off + (size_t)frame * L.blur_bytes;
    const int step = L.stride;
    const uint8_t* center = img + (size_t)kp.y * step + kp.x;
    unsigned nib = 0;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const signed char* p = &c_pattern[(lane * 4 + t) * 4];
        float x0 = (float)p[0], y0 = (float)p[1], x1 = (float)p[2], y1 = (float)p[3];
        int r0 = slamit_round_f(x0 * b + y0 * a), c0 = slamit_round_f(x0 * a - y0 * b);
        int r1 = slamit_round_f(x1 * b + y1 * a), c1 = slamit_round_f(x1 * a - y1 * b);
        int t0 = center[r0 * step + c0], t1 = center[r1 * step + c1];
        nib |= (unsigned)(t0 < t1) << t;
    }
    unsigned other = __shfl_xor(nib, 1, WAVE);
    if ((lane & 1) == 0)
        out_desc[((size_t)frame * out_cap + o) * SLAMIT_DESC_BYTES + (lane >> 1)] = (uint8_t)(nib | (other << 4));
    if (lane == 0) {
        slamit_kp k;
        float fx = (float)kp.x, fy = (float)kp.y;
        if (level != 0) { fx *= L.scale; fy *= L.scale; }  // ORBextractor.cc:1126-1132
        k.x = fx; k.y = fy; k.size = L.patch_size; k.angle = kp.angle; k.response = kp.response;
        k.octave = level; k.class_id = -1;
        out_kps[(size_t)frame * out_cap + o] = k;
    }
}

// --------------------------------------------------------------------------------------------
// copyMakeBorder(REFLECT_101, 19 px): only materialised when the caller asks for
// mvImagePyramid[level] (slamit_orb_level); the extractor itself never reads the border.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pad_level_kernel(const uint8_t* __restrict__ src, int w, int h,
                                                        size_t sstride, uint8_t* __restrict__ dst) {
    const int B = SLAMIT_EDGE_THRESHOLD;
    const int pw = w + 2 * B, ph = h + 2 * B;
    int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= pw || y >= ph) return;
    dst[(size_t)y * pw + x] = src[(size_t)reflect101(y - B, h) * sstride + reflect101(x - B, w)];
}

// candidates of one (frame, level) as (x, y, score) int triplets for the debug getter
__global__ void decode_candidates_kernel(const OrbLevel* __restrict__ levels, int level,
                                         const unsigned long long* __restrict__ K, int n,
                                         unsigned long long* __restrict__ order_out, int* __restrict__ xys) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const OrbLevel& L = levels[level];
    unsigned long long c = K[k];
    unsigned order = (unsigned)c;
    int lx = order & 63, ly = (order >> 6) & 63, cell = order >> 12;
    int ci = cell / L.nCols, cj = cell - ci * L.nCols;
    xys[3 * k + 0] = cj * L.wCell + lx;
    xys[3 * k + 1] = ci * L.hCell + ly;
    xys[3 * k + 2] = (int)(c >> 32);
    order_out[k] = order;
}

// --------------------------------------------------------------------------------------------
// launch wrappers (called from orb_api.hip)
// --------------------------------------------------------------------------------------------
extern "C++" {

hipError_t orbk_upload_pattern(hipStream_t st) {
    return hipMemcpyToSymbolAsync(HIP_SYMBOL(c_pattern), slamit_orb_pattern, SLAMIT_ORB_PATTERN_INTS, 0,
                                  hipMemcpyHostToDevice, st);
}

void orbk_resize(hipStream_t st, const uint8_t* src, int sw, int sh, size_t sstride, size_t sframe,
                 uint8_t* dst, int dw, int dh, size_t dstride, size_t dframe, const int* xofs,
                 const short* ialpha, const int* yofs, const short* ibeta, int nframes) {
    dim3 grid((dw + 255) / 256, (dh + 3) / 4, nframes);
    hipLaunchKernelGGL(resize_level_kernel, grid, dim3(256), 0, st, src, sw, sh, sstride, sframe, dst, dw,
                       dh, dstride, dframe, xofs, ialpha, yofs, ibeta);
}

void orbk_fast(hipStream_t st, const OrbLevel* levels, int nlevels, int cells_per_frame,
               const uint8_t* img0, size_t img0_stride, size_t img0_frame, const uint8_t* pyr,
               unsigned long long* cand, size_t cand_frame_stride, int* cand_count, int iniTh, int minTh,
               int nframes) {
    hipLaunchKernelGGL(fast_cells_kernel, dim3(cells_per_frame, nframes), dim3(256), 0, st, levels, nlevels,
                       img0, img0_stride, img0_frame, pyr, cand, cand_frame_stride, cand_count, iniTh, minTh);
}

size_t orbk_octree_smem(int node_cap) {
    return (size_t)node_cap * (8 + 2 * 8 + 2 * 4 + 4 * 4 + 5 * 4 + 4 + 4) + 64;
}

hipError_t orbk_octree_prepare(int node_cap) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(octree_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)orbk_octree_smem(node_cap));
}

void orbk_octree(hipStream_t st, const OrbLevel* levels, int nlevels, const unsigned long long* cand,
                 size_t cand_frame_stride, const int* cand_count, uint32_t* ws_xy, uint16_t* ws_node,
                 OrbLevelKp* lkp, size_t kp_frame_stride, int* kp_count, int node_cap, int nframes,
                 int level_override) {
    dim3 grid(level_override >= 0 ? 1 : nlevels, nframes);
    hipLaunchKernelGGL(octree_kernel, grid, dim3(256), orbk_octree_smem(node_cap), st, levels, nlevels, cand,
                       cand_frame_stride, cand_count, ws_xy, ws_node, lkp, kp_frame_stride, kp_count, node_cap,
                       level_override);
}

void orbk_ic_angle(hipStream_t st, const OrbLevel* levels, int nlevels, const uint8_t* img0,
                   size_t img0_stride, size_t img0_frame, const uint8_t* pyr, OrbLevelKp* lkp,
                   size_t kp_frame_stride, const int* kp_count, int max_kp, int nframes) {
    hipLaunchKernelGGL(ic_angle_kernel, dim3((max_kp + 3) / 4, nlevels, nframes), dim3(256), 0, st, levels,
                       nlevels, img0, img0_stride, img0_frame, pyr, lkp, kp_frame_stride, kp_count);
}

void orbk_blur(hipStream_t st, const uint8_t* src, int w, int h, size_t sstride, size_t sframe, uint8_t* dst,
               size_t dstride, size_t dframe, int nframes) {
    hipLaunchKernelGGL(blur_level_kernel, dim3((w + 63) / 64, (h + 15) / 16, nframes), dim3(256), 0, st, src, w,
                       h, sstride, sframe, dst, dstride, dframe);
}

void orbk_describe(hipStream_t st, const OrbLevel* levels, int nlevels, const uint8_t* blur,
                   const OrbLevelKp* lkp, size_t kp_frame_stride, const int* kp_count, slamit_kp* out_kps,
                   uint8_t* out_desc, int out_cap, int* out_n, int max_kp, int nframes) {
    hipLaunchKernelGGL(describe_kernel, dim3((max_kp + 3) / 4, nlevels, nframes), dim3(256), 0, st, levels,
                       nlevels, blur, lkp, kp_frame_stride, kp_count, out_kps, out_desc, out_cap, out_n);
}

void orbk_pad(hipStream_t st, const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst) {
    hipLaunchKernelGGL(pad_level_kernel, dim3((w + 38 + 63) / 64, (h + 38 + 3) / 4), dim3(256), 0, st, src, w, h,
                       sstride, dst);
}

void orbk_decode_candidates(hipStream_t st, const OrbLevel* levels, int level, const unsigned long long* K,
                            int n, unsigned long long* order_out, int* xys) {
    if (n <= 0) return;
    hipLaunchKernelGGL(decode_candidates_kernel, dim3((n + 255) / 256), dim3(256), 0, st, levels, level, K, n,
                       order_out, xys);
}

}  // extern "C++"
