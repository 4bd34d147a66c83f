// orb_fast_strip.hip — FAST-9/16 at iniThFAST as a strip-walking wavefront (round 2 rewrite of K2).
//
// Replaces the first attempt of the reference's per-cell loop  cv::FAST(cell, iniThFAST, true)
// (ORB_SLAM2/src/ORBextractor.cc:805-849) for every cell whose result is NOT empty; cells that stay empty are
// queued for the per-cell kernel of orb_kernels.hip, which runs their minThFAST retry (:829-833).
//
// Work unit ("job") = one cell ROW of one strip of <= 8 adjacent cells of one level of one frame, one wavefront:
//   * lane L owns the 4 pixels x0 + 4L .. +3 of every row (x0 is 4-byte aligned: ONE coalesced dword load per
//     lane and row, 256 contiguous bytes per wave instruction, the next block of 4 rows in flight in registers);
//   * rows go through a 16-row ring in LDS (kept twice at the wrap so that a ring pixel is "base + immediate");
//   * compass pre-test on 4 pixels per lane with byte-parallel v_lerp_u8 (a 9-arc of the 16-ring always holds two
//     adjacent compass points of one polarity).  The vertical tests are shared by the centre rows c and c + 3:
//     "S is darker than the centre" at row c IS "N is brighter than the centre" at row c + 3;
//   * survivors are compacted (ballot + mbcnt) into a ring list; 128 of them are scored per pass, TWO per lane on
//     packed 16-bit lanes: the 16 ring bytes of both pixels become f16 numbers 1024 + v (one v_lshl_or + one
//     v_xor, which also flips the values of a "darker" candidate so that only the "brighter" score is needed),
//     then 40 v_pk_minimum3_f16 / v_pk_maximum3_f16 give  max over the 16 arcs of (min over the 9 pixels);
//   * scores >= t go to a 16-row score ring; every 7 rows NMS runs byte-parallel over that ring (strict >,
//     neighbours of another cell or outside the scan area count as 0, as cv::FAST sees them) and the kept corners
//     are appended, as (score << 32 | order) keys, to the (frame, level) candidate list of the octree pass.
//
// Everything is integer / exact: the f16 lanes only ever hold the integers 1024 .. 1279 and are only compared.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "orb_types.h"

#define WAVE 64
#define FS_IMG_PITCH 260            // 65 dwords: consecutive rows start one bank apart
#define FS_IMG_SLOTS 22             // slot = (row & 15) + 3; rows with (row & 15) < 3 / >= 13 are also kept at slot +- 16
#define FS_SC_PITCH 264             // 4 bytes of pad on both sides of the 256 score bytes of a row
#define FS_LIST_CAP 512             // survivor ring list (u16 entries), power of two: < 128 left over + <= 256 of a row + 64
#define FS_KP_CAP 256               // kept corners waiting for their slot in the candidate list (u64); a row adds <= 192
#define FS_MAX_CELLS 8
#define FS_IMG_BYTES (4 + FS_IMG_SLOTS * FS_IMG_PITCH + 12)
#define FS_SC_BYTES (16 * FS_SC_PITCH)
#define FS_WAVE_LDS ((FS_IMG_BYTES + FS_SC_BYTES + 2 * FS_LIST_CAP + 8 * FS_KP_CAP + 4 * FS_MAX_CELLS + 15) & ~15)
#define FS_JOB_WORDS 8

// Job table entry (8 words), built by orbk_fast_strip_jobs:
//   0: level | ncells << 8 | ch << 16 | wCell << 24          (ch = rows of the cell windows incl. the 3 + 3 halo)
//   1: x0 | iniY << 16                                       (x0: first pixel of lane 0, multiple of 4)
//   2: sx0 | sx1 << 16                                       (scan pixels of the strip, relative to x0: [sx0, sx1))
//   3: index of the strip's first cell in the level (ci * nCols + cj)
//   4: index of the strip's first cell in the per-cell kernel's table (orbk_fast_cells)
//   5: floor(2^20 / wCell) + 1
//   6: last readable dword of a row, relative to x0, as a lane index (lanes beyond re-read it)
//   7: unused

__device__ __forceinline__ void fs_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ uint32_t fs_pkmin3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t fs_pkmax3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t fs_pkmax(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// any three-input boolean function, full rate on gfx950
#define FS_BITOP3(a, b, c, imm) __builtin_amdgcn_bitop3_b32((a), (b), (c), (imm))

struct FsCtx {
    uint8_t* img;                 // ring rows of the level image (wave private, LDS)
    uint8_t* sc;                  // score ring
    unsigned short* list;         // survivors: x | row << 8 | darker << 14
    unsigned long long* kp;       // kept corners
    int* cnt;                     // kept corners per cell of the strip
    int lane;
    int th;
    // survivor ring
    int head, count;
    // kept corner buffer
    int nkp;
    // candidate list of (frame, level)
    unsigned long long* out;
    int* out_count;
    int out_cap;
    // strip geometry
    int sx0, wCell, c0;
    unsigned inv_w;
    unsigned valid, edgeL, edgeR; // per-lane byte masks (0x80 per pixel)
};

// ---- scoring: <= 128 listed pixels per pass, two per lane -------------------------------------------------------
__device__ __forceinline__ void fs_score_pass(FsCtx& K, int n) {
    const int lane = K.lane;
    const bool vlo = lane < n, vhi = lane + 64 < n;
    // lanes without an entry re-score entry 0 of the pass (always present) and do not store
    const unsigned elo = K.list[(K.head + (vlo ? lane : 0)) & (FS_LIST_CAP - 1)];
    const unsigned ehi = K.list[(K.head + (vhi ? lane + 64 : 0)) & (FS_LIST_CAP - 1)];
    const unsigned xlo = elo & 255u, rlo = (elo >> 8) & 63u, xhi = ehi & 255u, rhi = (ehi >> 8) & 63u;
    // top-left corner of the 7x7 neighbourhood: slot (row & 15) + 3 - 3, column x - 3
    const uint8_t* plo = K.img + 4 + __umul24(rlo & 15u, FS_IMG_PITCH) + xlo - 3u;
    const uint8_t* phi = K.img + 4 + __umul24(rhi & 15u, FS_IMG_PITCH) + xhi - 3u;
    // 1024 + v as f16 in both halves; a darker-than-centre candidate is scored on 255 - v
    const uint32_t flip = 0x64006400u ^ ((elo >> 14) & 1u ? 0xFFu : 0u) ^ ((ehi >> 14) & 1u ? 0xFF0000u : 0u);
#define FS_PX(dx, dy) ((((uint32_t)phi[((dy) + 3) * FS_IMG_PITCH + (dx) + 3] << 16) | (uint32_t)plo[((dy) + 3) * FS_IMG_PITCH + (dx) + 3]) ^ flip)
    // ring in cv::FAST's order: (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
    uint32_t r[16];
    r[0] = FS_PX(0, 3);    r[1] = FS_PX(1, 3);    r[2] = FS_PX(2, 2);    r[3] = FS_PX(3, 1);
    r[4] = FS_PX(3, 0);    r[5] = FS_PX(3, -1);   r[6] = FS_PX(2, -2);   r[7] = FS_PX(1, -3);
    r[8] = FS_PX(0, -3);   r[9] = FS_PX(-1, -3);  r[10] = FS_PX(-2, -2); r[11] = FS_PX(-3, -1);
    r[12] = FS_PX(-3, 0);  r[13] = FS_PX(-3, 1);  r[14] = FS_PX(-2, 2);  r[15] = FS_PX(-1, 3);
    const uint32_t ctr = FS_PX(0, 0);
#undef FS_PX
    uint32_t lo3[16], l9[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) lo3[k] = fs_pkmin3(r[k], r[(k + 1) & 15], r[(k + 2) & 15]);
#pragma unroll
    for (int k = 0; k < 16; ++k) l9[k] = fs_pkmin3(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
    const uint32_t m = fs_pkmax(fs_pkmax3(fs_pkmax3(l9[0], l9[1], l9[2]), fs_pkmax3(l9[3], l9[4], l9[5]), fs_pkmax3(l9[6], l9[7], l9[8])),
                                fs_pkmax3(fs_pkmax3(l9[9], l9[10], l9[11]), fs_pkmax3(l9[12], l9[13], l9[14]), l9[15]));
    // score = (best arc minimum) - centre - 1 in the (possibly flipped) value domain; < t means "no corner at t"
    const int slo = (int)(m & 255u) - (int)(ctr & 255u) - 1;
    const int shi = (int)((m >> 16) & 255u) - (int)((ctr >> 16) & 255u) - 1;
    if (vlo && slo >= K.th) K.sc[__umul24(rlo & 15u, FS_SC_PITCH) + 4u + xlo] = (uint8_t)slo;
    if (vhi && shi >= K.th) K.sc[__umul24(rhi & 15u, FS_SC_PITCH) + 4u + xhi] = (uint8_t)shi;
    K.head = (K.head + n) & (FS_LIST_CAP - 1);
    K.count -= n;
}

// appends the pixels flagged in `flags` (MSB of each byte) of tile row `row`; darkonly marks darker-than-centre candidates.
// The caller guarantees room for 4 x 64 entries.
__device__ __forceinline__ void fs_append(FsCtx& K, unsigned flags, unsigned darkonly, int row) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool pass = (flags >> (8 * j + 7)) & 1u;
        const unsigned long long mk = __ballot(pass);
        if (mk) {
            const int pos = K.count + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
            if (pass) K.list[(K.head + pos) & (FS_LIST_CAP - 1)] =
                (unsigned short)((unsigned)(4 * K.lane + j) | ((unsigned)row << 8) | (((darkonly >> (8 * j + 7)) & 1u) << 14));
            K.count += __popcll(mk);
        }
    }
}

// flushes the kept corners to the (frame, level) candidate list
__device__ __forceinline__ void fs_flush(FsCtx& K) {
    fs_wave_sync();
    if (K.nkp > 0) {
        int base = 0;
        if (K.lane == 0) base = atomicAdd(K.out_count, K.nkp);
        base = __builtin_amdgcn_readfirstlane(base);
        for (int i = K.lane; i < K.nkp; i += WAVE) {
            const int o = base + i;
            if (o < K.out_cap) K.out[o] = K.kp[i];
        }
    }
    K.nkp = 0;
    fs_wave_sync();
}

// NMS over the scan rows [y0, y1] of the score ring (all their neighbours are final) + emission of the kept corners
__device__ __forceinline__ void fs_nms(FsCtx& K, int y0, int y1) {
    const int lane = K.lane;
    const uint8_t* base = K.sc + 4 + 4 * lane;
#define FS_ROW3(y, c, l, r) do { const uint8_t* p_ = base + __umul24((unsigned)((y) & 15), FS_SC_PITCH); \
        const uint32_t c_ = *reinterpret_cast<const uint32_t*>(p_), l_ = *reinterpret_cast<const uint32_t*>(p_ - 4), r_ = *reinterpret_cast<const uint32_t*>(p_ + 4); \
        c = c_; l = __builtin_amdgcn_alignbyte(c_, l_, 3); r = __builtin_amdgcn_alignbyte(r_, c_, 1); } while (0)
    uint32_t uc, ul, ur, cc, cl, cr, dc, dl, dr;
    FS_ROW3(y0 - 1, uc, ul, ur);
    FS_ROW3(y0, cc, cl, cr);
    for (int y = y0; y <= y1; ++y) {
        FS_ROW3(y + 1, dc, dl, dr);
        // n >= s per byte  <=>  MSB of (n + ~s + 1) >> 1
        const uint32_t ns = ~cc, one = 0x01010101u;
        const uint32_t gl = FS_BITOP3(__builtin_amdgcn_lerp(ul, ns, one), __builtin_amdgcn_lerp(cl, ns, one), __builtin_amdgcn_lerp(dl, ns, one), 0xFE);
        const uint32_t gr = FS_BITOP3(__builtin_amdgcn_lerp(ur, ns, one), __builtin_amdgcn_lerp(cr, ns, one), __builtin_amdgcn_lerp(dr, ns, one), 0xFE);
        const uint32_t gv = __builtin_amdgcn_lerp(uc, ns, one) | __builtin_amdgcn_lerp(dc, ns, one);
        // a neighbour in the next cell is outside this cell's FAST image: it counts as 0 (never >= a corner's score)
        const uint32_t beaten = gv | (gl & ~K.edgeL) | (gr & ~K.edgeR);
        const uint32_t keep = ~beaten & K.valid & 0x80808080u;     // a pixel with score 0 is always beaten by uc >= 0
        const int cnt = __popc(keep);
        const unsigned long long b1 = __ballot(cnt >= 1);
        if (b1) {
            // two strict maxima of ONE cell are never adjacent, but the last column of a cell and the first column of the next
            // one may both keep their pixel: up to 3 per dword (a dword holds at most one cell boundary)
            const unsigned long long b2 = __ballot(cnt >= 2), b3 = b2 ? __ballot(cnt >= 3) : 0ull;
            if (K.nkp > FS_KP_CAP - 192) fs_flush(K);
            int pos = K.nkp + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, 0u)) +
                      (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b2, 0u));
            if (b3) pos += (int)__builtin_amdgcn_mbcnt_hi((unsigned)(b3 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b3, 0u));
            unsigned kk = keep;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (q < cnt) {
                    const int j = (__ffs(kk) - 1) >> 3;
                    kk &= kk - 1u;
                    const unsigned S = (cc >> (8 * j)) & 255u;
                    const unsigned xr = (unsigned)(4 * lane + j - K.sx0);
                    const unsigned cx = __umul24(xr, K.inv_w) >> 20;
                    const unsigned wx = xr - __umul24(cx, (unsigned)K.wCell) + 3u;
                    const unsigned order = ((unsigned)(K.c0 + (int)cx) << 12) | ((unsigned)y << 6) | wx;
                    K.kp[pos + q] = ((unsigned long long)S << 32) | order;
                    atomicAdd(&K.cnt[cx], 1);
                }
            }
            K.nkp += __popcll(b1) + __popcll(b2) + __popcll(b3);
        }
        uc = cc; ul = cl; ur = cr; cc = dc; cl = dl; cr = dr;
    }
#undef FS_ROW3
}

__global__ __launch_bounds__(256) void fast_strip_kernel(
    const FastTab tab, const uint4* __restrict__ jobs, int njobs, int nlevels,
    const uint8_t* __restrict__ img0, unsigned img0_stride, size_t img0_frame, const uint8_t* __restrict__ pyr,
    unsigned long long* __restrict__ cand, size_t cand_frame_stride, int* __restrict__ cand_count,
    uint32_t* __restrict__ fb_list, int* __restrict__ fb_count, int th) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fs_lds[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int job = blockIdx.x * 4 + wv;
    const int frame = blockIdx.y;
    if (job >= njobs) return;
    const uint4 ja = jobs[2 * job], jb = jobs[2 * job + 1];
    const int level = (int)(ja.x & 255u), ncell = (int)((ja.x >> 8) & 255u), ch = (int)((ja.x >> 16) & 255u), wCell = (int)(ja.x >> 24);
    const int x0 = (int)(ja.y & 0xFFFFu), iniY = (int)(ja.y >> 16);
    const int sx0 = (int)(ja.z & 0xFFFFu), sx1 = (int)(ja.z >> 16);
    const FastLevel& L = tab.lv[level];

    uint8_t* wbase = fs_lds + wv * FS_WAVE_LDS;
    FsCtx K;
    K.img = wbase;
    K.sc = wbase + FS_IMG_BYTES;
    K.list = reinterpret_cast<unsigned short*>(K.sc + FS_SC_BYTES);
    K.kp = reinterpret_cast<unsigned long long*>(wbase + ((FS_IMG_BYTES + FS_SC_BYTES + 2 * FS_LIST_CAP + 7) & ~7));
    K.cnt = reinterpret_cast<int*>(reinterpret_cast<uint8_t*>(K.kp) + 8 * FS_KP_CAP);
    K.lane = lane; K.th = th; K.head = 0; K.count = 0; K.nkp = 0;
    K.out = cand + L.cand_off + (size_t)frame * cand_frame_stride;
    K.out_count = &cand_count[(frame * nlevels + level) * ORB_CC_PAD];
    K.out_cap = L.cand_cap;
    K.sx0 = sx0; K.wCell = wCell; K.c0 = (int)ja.w; K.inv_w = jb.y;
    {   // per-lane byte masks: scan pixels; first / last pixel of a cell
        unsigned valid = 0, eL = 0, eR = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = 4 * lane + j;
            if (x >= sx0 && x < sx1) {
                valid |= 0x80u << (8 * j);
                const unsigned xr = (unsigned)(x - sx0);
                const unsigned cx = __umul24(xr, K.inv_w) >> 20;
                const unsigned wx = xr - __umul24(cx, (unsigned)wCell);
                if (wx == 0) eL |= 0x80u << (8 * j);
                if (wx == (unsigned)(wCell - 1)) eR |= 0x80u << (8 * j);
            }
        }
        K.valid = valid; K.edgeL = eL; K.edgeR = eR;
    }
    if (lane < FS_MAX_CELLS) K.cnt[lane] = 0;

    const uint8_t* src;
    unsigned stride;
    if (level == 0) { src = img0 + (size_t)frame * img0_frame; stride = img0_stride; }
    else { src = pyr + L.plane_off + (size_t)frame * L.plane_bytes; stride = (unsigned)L.stride; }
    // lanes past the last dword that holds needed pixels re-read that dword (their pixels are never valid)
    const uint8_t* col = src + (size_t)iniY * stride + x0 + 4 * min(lane, (int)jb.z);

    // compass pre-test at threshold t (see orb_kernels.hip, K2): a - c > t  <=>  MSB of lerp(lerp(a, ~c, rnd), kc, 0)
    const int T = th + 256;
    const unsigned rnd = (T & 1) ? 0x01010101u : 0u;
    const unsigned kc = (unsigned)(256 - ((T + 1) >> 1)) * 0x01010101u;
#define FS_GT(a, nb) __builtin_amdgcn_lerp(__builtin_amdgcn_lerp((a), (nb), rnd), kc, 0u)   /* a > b + t, nb = ~b */

    // Rows enter in blocks of 4: the loads of block b + 1 are in flight while the rows of block b are processed.  (A
    // register queue rotated by v_mov would make every move wait for its load; static names need no moves.)
#define FS_LOAD(k) (*reinterpret_cast<const uint32_t*>(col + __umul24((unsigned)min((k), ch - 1), stride)))
    uint32_t t0 = FS_LOAD(0), t1 = FS_LOAD(1), t2 = FS_LOAD(2), t3 = FS_LOAD(3);
    uint32_t fb1 = 0, fb2 = 0, fb3 = 0, fd1 = 0, fd2 = 0, fd3 = 0;   // "row brighter / darker than the row 3 above" of the last 3 rows
    int nms_next = 3, since_drain = 0;
    uint32_t* imgw = reinterpret_cast<uint32_t*>(K.img + 4) + lane;              // lane's dword of ring slot 0
    uint32_t* scw = reinterpret_cast<uint32_t*>(K.sc + 4) + lane;
#define FS_STORE(row, v) do { if ((row) < ch) { const int rs_ = (row) & 15; imgw[(rs_ + 3) * (FS_IMG_PITCH / 4)] = (v); \
        if (rs_ < 3) imgw[(rs_ + 19) * (FS_IMG_PITCH / 4)] = (v); if (rs_ >= 13) imgw[(rs_ - 13) * (FS_IMG_PITCH / 4)] = (v); \
        scw[rs_ * (FS_SC_PITCH / 4)] = 0u; } } while (0)
    for (int rb = 0; rb < ch; rb += 4) {
        FS_STORE(rb, t0); FS_STORE(rb + 1, t1); FS_STORE(rb + 2, t2); FS_STORE(rb + 3, t3);
        t0 = FS_LOAD(rb + 4); t1 = FS_LOAD(rb + 5); t2 = FS_LOAD(rb + 6); t3 = FS_LOAD(rb + 7);
        fs_wave_sync();
        const int rend = min(rb + 4, ch);
        for (int r = rb; r < rend; ++r) {
            unsigned flags = 0, darkonly = 0, both = 0;
            uint32_t bS = 0, dS = 0;
            const int c = r - 3;
            if (r >= 3) {
                const uint32_t d = imgw[((r & 15) + 3) * (FS_IMG_PITCH / 4)];
                const uint32_t* cw = imgw + ((c & 15) + 3) * (FS_IMG_PITCH / 4);
                const uint32_t C = cw[0], nC = ~C, nd = ~d;
                bS = FS_GT(d, nC); dS = FS_GT(C, nd);          // row r brighter / darker than row r - 3
                if (r >= 6) {
                    const uint32_t Lw = cw[-1], Rw = cw[1];
                    const uint32_t E = __builtin_amdgcn_alignbyte(Rw, C, 3), Wv = __builtin_amdgcn_alignbyte(C, Lw, 1);
                    const uint32_t bE = FS_GT(E, nC), bW = FS_GT(Wv, nC), dE = FS_GT(C, ~E), dW = FS_GT(C, ~Wv);
                    // N tests of this centre row = S tests of three steps ago with the roles swapped
                    const uint32_t bN = fd3, dN = fb3;
                    const uint32_t pb = FS_BITOP3(bN | bS, bE, bW, 0xE0) & K.valid;   // a & (b | c)
                    const uint32_t pd = FS_BITOP3(dN | dS, dE, dW, 0xE0) & K.valid;
                    flags = pb | pd; darkonly = pd & ~pb; both = pb & pd & 0x80808080u;
                }
            }
            fb3 = fb2; fb2 = fb1; fb1 = bS;
            fd3 = fd2; fd2 = fd1; fd1 = dS;
            if (r >= 6) {
                // One site for every action on the survivor list: the row's pixels, then one entry as a darker candidate
                // for each pixel that passed both polarities, and a scoring pass whenever 128 entries wait (or, at the
                // end of a 7-row block, until the list is empty).
                ++since_drain;
                const bool block_end = since_drain == 7 || r == ch - 1;
                bool main_done = false;
                for (;;) {
                    const bool both_left = __ballot(both != 0u) != 0ull;
                    if (K.count >= 128 || (block_end && main_done && !both_left && K.count > 0)) {
                        fs_wave_sync();
                        fs_score_pass(K, min(K.count, 128));
                        fs_wave_sync();
                        continue;
                    }
                    if (!main_done) { fs_append(K, flags, darkonly, c); main_done = true; continue; }
                    if (both_left) {
                        const unsigned first = both & (0u - both);           // lowest flagged pixel of the lane
                        fs_append(K, first, first, c);
                        both &= both - 1u;
                        continue;
                    }
                    break;
                }
                if (block_end) {
                    since_drain = 0;
                    const int y1 = (r == ch - 1) ? c : c - 1;
                    if (y1 >= nms_next) fs_nms(K, nms_next, y1);
                    nms_next = y1 + 1;
                }
            }
        }
    }
#undef FS_LOAD
#undef FS_STORE
    fs_flush(K);
    // cells without a corner after NMS go to the per-cell kernel for the minThFAST retry
    if (lane < ncell && K.cnt[lane] == 0) {
        const int i = atomicAdd(fb_count, 1);
        fb_list[i] = ((unsigned)frame << 20) | (unsigned)((int)jb.x + lane);
    }
#undef FS_GT
}

// ---- host ----------------------------------------------------------------------------------------------------------
extern "C++" {

// Job table: for every level and cell row, the non-empty cells (same skip rules as orbk_fast_cells,
// ORBextractor.cc:810,819) are grouped left to right into strips whose windows fit 256 pixels from an aligned start.
int orbk_fast_strip_jobs(const OrbLevel* host_levels, int nlevels, std::vector<uint32_t>& out) {
    out.clear();
    int tab_index = 0;
    for (int l = 0; l < nlevels; ++l) {
        const OrbLevel& L = host_levels[l];
        for (int ci = 0; ci < L.nRows; ++ci) {
            const int iniY = ORB_MIN_BORDER + ci * L.hCell;
            if (iniY >= L.maxBorderY - 3) continue;
            const int ch = std::min(L.hCell + 6, L.maxBorderY - iniY);
            std::vector<int> cols;   // non-empty cells of this row
            for (int cj = 0; cj < L.nCols; ++cj) {
                const int iniX = ORB_MIN_BORDER + cj * L.wCell;
                if (iniX >= L.maxBorderX - 6) continue;
                const int cw = std::min(L.wCell + 6, L.maxBorderX - iniX);
                if (cw - 6 <= 0 || ch - 6 <= 0) continue;
                cols.push_back(cj);
            }
            size_t i = 0;
            while (i < cols.size()) {
                const int cj0 = cols[i];
                const int iniX0 = ORB_MIN_BORDER + cj0 * L.wCell, x0 = iniX0 & ~3;
                size_t n = 0;
                int xend = 0;
                while (i + n < cols.size() && n < FS_MAX_CELLS) {
                    const int cj = cols[i + n];
                    if (cj != cj0 + (int)n) break;   // cells of a strip are adjacent
                    const int iniX = ORB_MIN_BORDER + cj * L.wCell, cw = std::min(L.wCell + 6, L.maxBorderX - iniX);
                    if (iniX + cw - x0 > 256) break;
                    xend = iniX + cw;
                    ++n;
                }
                if (n == 0) return -1;   // a single cell wider than a strip: not a geometry this kernel takes
                const int sx0 = iniX0 + 3 - x0, sx1 = xend - 3 - x0;
                const int last_lane = std::min(63, ((xend - 1) - x0) >> 2);
                const uint32_t w[FS_JOB_WORDS] = {
                    (uint32_t)l | ((uint32_t)n << 8) | ((uint32_t)ch << 16) | ((uint32_t)L.wCell << 24),
                    (uint32_t)x0 | ((uint32_t)iniY << 16), (uint32_t)sx0 | ((uint32_t)sx1 << 16), (uint32_t)(ci * L.nCols + cj0),
                    (uint32_t)tab_index, (uint32_t)((1u << 20) / (unsigned)L.wCell + 1u), (uint32_t)last_lane, 0u};
                out.insert(out.end(), w, w + FS_JOB_WORDS);
                tab_index += (int)n;
                i += n;
            }
        }
    }
    return (int)(out.size() / FS_JOB_WORDS);
}

bool orbk_fast_strip_supported(const OrbLevel* host_levels, int nlevels) {
    for (int l = 0; l < nlevels; ++l) {
        const OrbLevel& L = host_levels[l];
        if (L.hCell + 6 > 64 || L.wCell + 6 + 3 > 256 || L.w >= 65536 || L.h >= 65536) return false;   // 6-bit tile rows
        if (L.ncells >= (1 << 20)) return false;
    }
    return true;
}

size_t orbk_fast_strip_smem() { return (size_t)4 * FS_WAVE_LDS; }

void orbk_fast_strip(hipStream_t st, const OrbLevel* host_levels, int nlevels, const uint32_t* d_jobs, int njobs,
                     const uint8_t* img0, size_t img0_stride, size_t img0_frame, const uint8_t* pyr,
                     unsigned long long* cand, size_t cand_frame_stride, int* cand_count, uint32_t* fb_list, int* fb_count,
                     int th, int nframes) {
    FastTab tab = {};
    for (int l = 0; l < nlevels && l < ORB_MAX_LEVELS; ++l) {
        const OrbLevel& S = host_levels[l];
        FastLevel& D = tab.lv[l];
        D.cell_base = S.cell_base; D.nCols = S.nCols; D.wCell = S.wCell; D.hCell = S.hCell;
        D.maxBorderX = S.maxBorderX; D.maxBorderY = S.maxBorderY; D.stride = S.stride; D.cand_cap = S.cand_cap;
        D.plane_off = S.plane_off; D.plane_bytes = S.plane_bytes; D.cand_off = S.cand_off;
    }
    hipLaunchKernelGGL(fast_strip_kernel, dim3((njobs + 3) / 4, nframes), dim3(256), orbk_fast_strip_smem(), st, tab,
                       reinterpret_cast<const uint4*>(d_jobs), njobs, nlevels, img0, (unsigned)img0_stride, img0_frame, pyr, cand,
                       cand_frame_stride, cand_count, fb_list, fb_count, th);
}

}  // extern "C++"
