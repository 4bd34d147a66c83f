// hamming.hip — 256-bit Hamming matching (include/slamit.h, slamit_hamming_*).
//
// Replaces ORBmatcher::DescriptorDistance (ORB_SLAM2/src/ORBmatcher.cc:1651-1667) and the
// best / second-best selection loops that call it (:85-117, :440-461, :1404-1428):
//     if (d < best) { second = best; best = d; idx = j; } else if (d < second) second = d;
// i.e. best = minimum distance with the FIRST index on ties, second = second smallest value of
// the multiset.  That pair is an associative reduction, so the train set is split over 4 lanes
// per query and merged with cross-lane shuffles:
//     merge((b1,i1,s1),(b2,i2,s2)) = (b1,i1,min(s1,b2)) if (b1,i1) < (b2,i2) else (b2,i2,min(s2,b1)).
// Workgroup = 256 threads = 64 queries x 4 train slices; train descriptors stream through LDS in
// tiles of 256 rows (8 KB), read as broadcast ds_read_b128; distance = 8 x (v_xor + v_bcnt).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "../../include/slamit.h"
#include "slamit_internal.h"

#define HM_TILE 256      // train rows per LDS tile: two uint4 per thread
#ifndef HM_SLICES
#define HM_SLICES 8     // lanes per query: each takes every HM_SLICES-th train row of a tile
#endif
#define HM_QPB (256 / HM_SLICES)   // queries per workgroup

// popcount(x) + acc in ONE instruction; left to itself the compiler builds a tree of 8 v_bcnt + 3 v_add3 per distance
__device__ __forceinline__ unsigned bcnt_acc(unsigned x, unsigned acc) {
    unsigned r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ unsigned umed3(unsigned a, unsigned b, unsigned c) {   // v_med3_u32 (no builtin for the unsigned form)
    unsigned r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__global__ __launch_bounds__(256) void hamming_best2_kernel(
    const uint8_t* __restrict__ q, const int* __restrict__ nq_arr, int nq_fixed, size_t q_stride,
    const uint8_t* __restrict__ t, const int* __restrict__ nt_arr, int nt_fixed, size_t t_stride,
    int* __restrict__ best_idx, int* __restrict__ best, int* __restrict__ second, size_t out_stride) {
    __shared__ uint4 tile[HM_TILE * 2];
    const int pair = blockIdx.y;
    const int nq = nq_arr ? nq_arr[pair] : nq_fixed;
    const int nt = nt_arr ? nt_arr[pair] : nt_fixed;
    const int tid = threadIdx.x;
    const int qi = blockIdx.x * HM_QPB + tid / HM_SLICES;
    const int slice = tid % HM_SLICES;
    if (blockIdx.x * HM_QPB >= nq) return;  // uniform
    const uint4* Q = reinterpret_cast<const uint4*>(q + (size_t)pair * q_stride);
    const uint4* T = reinterpret_cast<const uint4*>(t + (size_t)pair * t_stride);
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    const bool live = qi < nq;
    if (live) { a0 = Q[2 * (size_t)qi]; a1 = Q[2 * (size_t)qi + 1]; }
    // (best, second) are the two smallest keys (distance << 16 | train index): strict '<' with the first index
    // winning ties is exactly the order of these keys, and a median / min pair updates both without a branch
    unsigned kb = 0xFFFFFFFFu, ks = 0xFFFFFFFFu;
    // the next tile of train descriptors travels HBM -> registers while the current one is being scanned
    const uint4 zero = make_uint4(0, 0, 0, 0);
    // (Two named registers, not an array: the compiler moves a small array to LDS and then waits for the load at once.
    // Indices are clamped, not predicated: a select between a load and zero turns into a FLAT load, which also counts
    // in lgkmcnt and would make every LDS wait of the scan wait for the prefetch.  Rows >= nt of a tile are never read.)
    uint4 p0 = zero, p1 = zero;
    if (nt > 0) {
        const int last = 2 * min(nt, HM_TILE) - 1;
        p0 = T[min(tid, last)]; p1 = T[min(tid + 256, last)];
    }
    for (int base = 0; base < nt; base += HM_TILE) {
        const int rows = min(HM_TILE, nt - base);
        __syncthreads();
        tile[tid] = p0; tile[tid + 256] = p1;
        __syncthreads();
        if (base + HM_TILE < nt) {
            const int last = 2 * min(HM_TILE, nt - base - HM_TILE) - 1;
            const uint4* Tn = T + 2 * (size_t)(base + HM_TILE);
            p0 = Tn[min(tid, last)]; p1 = Tn[min(tid + 256, last)];
        }
#define HM_DIST(T0, T1) bcnt_acc(a1.w ^ T1.w, bcnt_acc(a1.z ^ T1.z, bcnt_acc(a1.y ^ T1.y, bcnt_acc(a1.x ^ T1.x, \
                        bcnt_acc(a0.w ^ T0.w, bcnt_acc(a0.z ^ T0.z, bcnt_acc(a0.y ^ T0.y, bcnt_acc(a0.x ^ T0.x, 0u))))))))
#define HM_TAKE(D, J) do { const unsigned k = ((D) << 16) | (unsigned)(base + (J)); \
            ks = umed3(kb, ks, k);   /* kb <= ks: the median of the three is the new second smallest */ \
            kb = min(kb, k); } while (0)
        int j = slice;
        // four rows per trip, their eight LDS reads issued before the first popcount (unrolled by hand: the asm in umed3
        // stops the unroller)
        for (; j + 3 * HM_SLICES < rows; j += 4 * HM_SLICES) {
            const uint4 t00 = tile[2 * j], t01 = tile[2 * j + 1], t10 = tile[2 * (j + HM_SLICES)], t11 = tile[2 * (j + HM_SLICES) + 1];
            const uint4 t20 = tile[2 * (j + 2 * HM_SLICES)], t21 = tile[2 * (j + 2 * HM_SLICES) + 1];
            const uint4 t30 = tile[2 * (j + 3 * HM_SLICES)], t31 = tile[2 * (j + 3 * HM_SLICES) + 1];
            const unsigned d0 = HM_DIST(t00, t01), d1 = HM_DIST(t10, t11), d2 = HM_DIST(t20, t21), d3 = HM_DIST(t30, t31);
            HM_TAKE(d0, j); HM_TAKE(d1, j + HM_SLICES); HM_TAKE(d2, j + 2 * HM_SLICES); HM_TAKE(d3, j + 3 * HM_SLICES);
        }
        for (; j < rows; j += HM_SLICES) { const uint4 t0 = tile[2 * j], t1 = tile[2 * j + 1]; const unsigned d = HM_DIST(t0, t1); HM_TAKE(d, j); }
#undef HM_DIST
#undef HM_TAKE
    }
    // merge the slices of each query (adjacent lanes): two smallest of the union
#pragma unroll
    for (int m = 1; m < HM_SLICES; m <<= 1) {
        const unsigned ob = (unsigned)__shfl_xor((int)kb, m, 64), os = (unsigned)__shfl_xor((int)ks, m, 64);
        ks = min(min(ks, os), max(kb, ob));
        kb = min(kb, ob);
    }
    const int b = kb == 0xFFFFFFFFu ? 256 : (int)(kb >> 16), bi = kb == 0xFFFFFFFFu ? -1 : (int)(kb & 0xFFFFu);
    const int s = ks == 0xFFFFFFFFu ? 256 : (int)(ks >> 16);
    if (live && slice == 0) {
        size_t o = (size_t)pair * out_stride + qi;
        best_idx[o] = bi; best[o] = b; second[o] = s;
    }
}

// ---- the same reduction on the matrix cores -------------------------------------------------------------------
// 256-bit Hamming distance as an int8 dot product: a descriptor bit becomes the byte +-64 (query: 1 -> -64, train:
// 1 -> +64), so a byte product is -4096 where the bits agree and +4096 where they differ and the sum over the 256
// positions is 4096 * (2 d - 256) = 8192 d - 2^20 -- exact in i32.  The 13 low bits of that sum are zero, so the train
// index j (< 8192) rides in the accumulator's initial value: v_mfma_i32_32x32x32_i8 leaves  8192 d - 2^20 + j  = the
// (distance, first index) key itself, and the epilogue per distance is a min and a median (2 VALU ops instead of the
// 19 of the xor / popcount kernel).  Unpacking costs one shift + one v_bitop3 per four positions:
//     byte = ((w << (7 - s)) & 0x80) | 0x40            (query; train uses ~w)
// Position order inside the K = 256 axis is whatever (lane half, shift class s, dword) gives -- the same for both operands.
// A = train tile (32 rows), B = queries (32 columns): a lane's 16 accumulators are 16 train rows of ONE query, so the
// running (best, second) keys are two registers per lane and query tile.  Workgroup = 4 wavefronts that share 64 queries
// (two 32-column tiles kept unpacked in 64 VGPRs) and take every 4th train tile; they merge through LDS at the end.
typedef int hm_v4i __attribute__((ext_vector_type(4)));
typedef int hm_v16i __attribute__((ext_vector_type(16)));
#define HMM_MAX_TRAIN 8192
#define HMM_NONE 0x40000000          // accumulator start of a train row >= nt: its key stays above every real one
#define HMM_EMPTY 0x7F000000         // "no key yet"

__device__ __forceinline__ hm_v4i hm_unpack_q(const uint4 w, const int s) {   // bit 1 -> 0xC0 (-64), bit 0 -> 0x40 (+64)
    const unsigned hi = 0x80808080u, mid = 0x40404040u;
    hm_v4i r;
    r.x = (int)__builtin_amdgcn_bitop3_b32(w.x << (7 - s), hi, mid, 0xEA);
    r.y = (int)__builtin_amdgcn_bitop3_b32(w.y << (7 - s), hi, mid, 0xEA);
    r.z = (int)__builtin_amdgcn_bitop3_b32(w.z << (7 - s), hi, mid, 0xEA);
    r.w = (int)__builtin_amdgcn_bitop3_b32(w.w << (7 - s), hi, mid, 0xEA);
    return r;
}
__device__ __forceinline__ hm_v4i hm_unpack_t(const uint4 w, const int s) {   // bit 1 -> 0x40 (+64), bit 0 -> 0xC0 (-64)
    const unsigned hi = 0x80808080u, mid = 0x40404040u;
    hm_v4i r;
    r.x = (int)__builtin_amdgcn_bitop3_b32(w.x << (7 - s), hi, mid, 0xAE);
    r.y = (int)__builtin_amdgcn_bitop3_b32(w.y << (7 - s), hi, mid, 0xAE);
    r.z = (int)__builtin_amdgcn_bitop3_b32(w.z << (7 - s), hi, mid, 0xAE);
    r.w = (int)__builtin_amdgcn_bitop3_b32(w.w << (7 - s), hi, mid, 0xAE);
    return r;
}
// v_med3_i32 has no builtin.  The compiler pads the wait states between an MFMA and a VALU instruction that reads its
// result only for instructions it knows, not inside asm: `after` is a value computed FROM the same MFMA result by an
// ordinary instruction (the v_min of the same key), which orders this asm behind that instruction and its padding.
__device__ __forceinline__ int imed3_after(int a, int b, int c, int after) {
    int r;
    asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c), "v"(after));
    return r;
}

// The kernel is bound by vector-instruction ISSUE, which the MFMAs share (8 of an MFMA's 32 cycles): per 64 queries x
// 32 train rows that is 16 MFMAs (128 cycles of issue) + 64 epilogue instructions + the train tile's unpacking (60,
// shared by all the query tiles of the wave) at ~4 cycles each.  A wave therefore carries FOUR query tiles (128
// queries, 128 VGPRs of unpacked queries + 64 of accumulators -> 2 waves per SIMD): 30 + 64 instructions per 16 MFMAs.
// Workgroup = 4 wavefronts = 4 train slices (a slice takes every 4th train tile) of the same 128 queries, one wave per
// SIMD; they merge through LDS at the end.  Measured (tools/diag/ubench/mfma_i8_rate.hip): a wave's
// MFMAs (16 ns each per SIMD) and its ~190 vector instructions per tile (~2 ns each) ADD UP -- 0.51 + 0.38 us per tile of 128
// queries x 32 train rows, the two waves of a SIMD running in step; starting one of them half a period late (s_sleep) or
// interleaving the epilogue of one half of the query tiles with the MFMAs of the other inside the wave (which doubles the
// unpacking) both left the time unchanged, so neither is kept.
#define HMM_WAVES 4                   // train slices = wavefronts per workgroup
#define HMM_NT 4                      // query tiles (32 queries each) per wavefront

__device__ __forceinline__ void hm_merge2(int& kb, int& ks, int ob, int os) {   // two smallest of the union of two (best, second) pairs
    ks = min(min(ks, os), max(kb, ob));
    kb = min(kb, ob);
}

__global__ __launch_bounds__(64 * HMM_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void hamming_best2_mfma_kernel(
    const uint8_t* __restrict__ q, const int* __restrict__ nq_arr, int nq_fixed, size_t q_stride,
    const uint8_t* __restrict__ t, const int* __restrict__ nt_arr, int nt_fixed, size_t t_stride,
    int* __restrict__ best_idx, int* __restrict__ best, int* __restrict__ second, size_t out_stride) {
    __shared__ int s_keys[HMM_WAVES - 1][64][2 * HMM_NT];       // slices 1..: (kb, ks) per query tile and lane
    const int pair = blockIdx.y;
    const int nq = __builtin_amdgcn_readfirstlane(nq_arr ? nq_arr[pair] : nq_fixed);
    const int nt = __builtin_amdgcn_readfirstlane(min(nt_arr ? nt_arr[pair] : nt_fixed, HMM_MAX_TRAIN));
    const int qbase = blockIdx.x * 32 * HMM_NT;
    if (qbase >= nq) return;  // uniform over the workgroup
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int col = lane & 31, h = lane >> 5;
    const uint4* Q = reinterpret_cast<const uint4*>(q + (size_t)pair * q_stride);
    const uint4* T = reinterpret_cast<const uint4*>(t + (size_t)pair * t_stride);
    // this lane's half (dwords 4h .. 4h+3) of its queries, unpacked once: 8 k-steps x 4 VGPRs per query tile
    hm_v4i bq[HMM_NT][8];
#pragma unroll
    for (int u = 0; u < HMM_NT; ++u) {
        const uint4 w = Q[2 * (size_t)min(qbase + 32 * u + col, nq - 1) + h];
#pragma unroll
        for (int s = 0; s < 8; ++s) bq[u][s] = hm_unpack_q(w, s);
    }
    // Accumulator start = row of (register r, lane half h) inside the tile: (r & 3) + 8 (r >> 2) + 4 h -- the same 16
    // registers for every tile.  The running keys are kept RELATIVE to the current tile's first row (stepping to the next
    // tile subtracts the stride from them) and the last tile's base is added back at the end.
    hm_v16i cinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = 4 * h + (r & 3) + 8 * (r >> 2);
    int kb[HMM_NT], ks[HMM_NT];
#pragma unroll
    for (int u = 0; u < HMM_NT; ++u) kb[u] = ks[u] = HMM_EMPTY;   // (room for the +- tile bases)
    const int ntiles = (nt + 31) >> 5;
    uint4 wt = make_uint4(0u, 0u, 0u, 0u);
    if (wv < ntiles) wt = T[2 * (size_t)min(wv * 32 + col, nt - 1) + h];
    int tile = wv;
    for (; tile < ntiles; tile += HMM_WAVES) {
        const uint4 w = wt;
        if (tile + HMM_WAVES < ntiles) wt = T[2 * (size_t)min((tile + HMM_WAVES) * 32 + col, nt - 1) + h];   // travels during this tile
        hm_v16i c[HMM_NT];
        {
            const hm_v4i a = hm_unpack_t(w, 0);
            hm_v16i cm = cinit;
            if (tile * 32 + 32 > nt) {   // the set's last, partial tile: rows >= nt start above every real key
#pragma unroll
                for (int r = 0; r < 16; ++r) cm[r] = tile * 32 + cinit[r] < nt ? cinit[r] : HMM_NONE;
            }
#pragma unroll
            for (int u = 0; u < HMM_NT; ++u) c[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[u][0], cm, 0, 0, 0);
        }
#pragma unroll
        for (int s = 1; s < 8; ++s) {
            const hm_v4i a = hm_unpack_t(w, s);
#pragma unroll
            for (int u = 0; u < HMM_NT; ++u) c[u] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bq[u][s], c[u], 0, 0, 0);
        }
        if (tile != wv) {
#pragma unroll
            for (int u = 0; u < HMM_NT; ++u) { kb[u] -= 32 * HMM_WAVES; ks[u] -= 32 * HMM_WAVES; }
        }
#pragma unroll
        for (int u = 0; u < HMM_NT; ++u) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = min(kb[u], c[u][r]);
                ks[u] = imed3_after(kb[u], ks[u], c[u][r], n);   // kb <= ks: the median of the three is the new second smallest
                kb[u] = n;
            }
        }
    }
    if (tile != wv) {   // this wave saw at least one tile: back to absolute train indices (base of its last tile)
        const int base = (tile - HMM_WAVES) * 32;
#pragma unroll
        for (int u = 0; u < HMM_NT; ++u) { kb[u] += base; ks[u] += base; }
    }
    // the two lane halves hold different train rows of the same query
#pragma unroll
    for (int u = 0; u < HMM_NT; ++u) hm_merge2(kb[u], ks[u], __shfl_xor(kb[u], 32, 64), __shfl_xor(ks[u], 32, 64));
    if (wv > 0) {
#pragma unroll
        for (int u = 0; u < HMM_NT; ++u) { s_keys[wv - 1][lane][2 * u] = kb[u]; s_keys[wv - 1][lane][2 * u + 1] = ks[u]; }
    }
    __syncthreads();
    if (wv != 0 || h != 0) return;
#pragma unroll
    for (int o = 0; o < HMM_WAVES - 1; ++o) {
#pragma unroll
        for (int u = 0; u < HMM_NT; ++u) hm_merge2(kb[u], ks[u], s_keys[o][lane][2 * u], s_keys[o][lane][2 * u + 1]);
    }
    // key = 8192 d - 2^20 + j; anything from HMM_NONE / the empty start is far above 2^29
#pragma unroll
    for (int u = 0; u < HMM_NT; ++u) {
        const int qi = qbase + 32 * u + col;
        if (qi >= nq) continue;
        const bool hb = kb[u] < 0x20000000, hs = ks[u] < 0x20000000;
        const size_t o = (size_t)pair * out_stride + qi;
        best_idx[o] = hb ? ((kb[u] + (1 << 20)) & 8191) : -1;
        best[o] = hb ? ((kb[u] + (1 << 20)) >> 13) : 256;
        second[o] = hs ? ((ks[u] + (1 << 20)) >> 13) : 256;
    }
}

__global__ __launch_bounds__(256) void hamming_matrix_kernel(const uint8_t* __restrict__ q, int nq,
                                                             const uint8_t* __restrict__ t, int nt,
                                                             uint16_t* __restrict__ out) {
    __shared__ uint4 tile[64 * 2];
    const int tid = threadIdx.x;
    const int j0 = blockIdx.x * 64, i0 = blockIdx.y * 64;
    const uint4* Q = reinterpret_cast<const uint4*>(q);
    const uint4* T = reinterpret_cast<const uint4*>(t);
    const int rows = min(64, nt - j0);
    for (int i = tid; i < rows * 2; i += 256) tile[i] = T[2 * (size_t)j0 + i];
    __syncthreads();
    const int j = tid & 63;
    for (int ii = tid >> 6; ii < 64; ii += 4) {
        int i = i0 + ii;
        if (i >= nq || j >= rows) continue;
        uint4 a0 = Q[2 * (size_t)i], a1 = Q[2 * (size_t)i + 1];
        uint4 t0 = tile[2 * j], t1 = tile[2 * j + 1];
        int d = __popc(a0.x ^ t0.x) + __popc(a0.y ^ t0.y) + __popc(a0.z ^ t0.z) + __popc(a0.w ^ t0.w) +
                __popc(a1.x ^ t1.x) + __popc(a1.y ^ t1.y) + __popc(a1.z ^ t1.z) + __popc(a1.w ^ t1.w);
        out[(size_t)i * nt + j0 + j] = (uint16_t)d;
    }
}

// MapPoint::ComputeDistinctiveDescriptors: one wavefront per map point.  Descriptors and the N x N
// distance matrix (u16) sit in LDS; each lane owns rows lane, lane+64 and finds its row's median by
// bisection on the value range [0, 256] (count of entries <= mid), then the wave takes the
// lexicographic minimum of (median, row) so the first least-median row wins like the reference's loop.
__global__ __launch_bounds__(64) void distinctive_kernel(const uint8_t* __restrict__ desc, const int* __restrict__ offsets,
                                                         int* __restrict__ best_idx, int* __restrict__ best_median) {
    __shared__ uint4 rowsd[SLAMIT_DISTINCTIVE_MAX * 2];
    __shared__ unsigned short D[SLAMIT_DISTINCTIVE_MAX * SLAMIT_DISTINCTIVE_MAX];
    const int p = blockIdx.x, lane = threadIdx.x;
    const int o = offsets[p], n = offsets[p + 1] - o;
    if (n <= 0) { if (lane == 0) { best_idx[p] = -1; best_median[p] = 0; } return; }
    const uint4* G = reinterpret_cast<const uint4*>(desc + (size_t)o * 32);
    for (int i = lane; i < 2 * n; i += 64) rowsd[i] = G[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n; i += 64) {
        const uint4 a0 = rowsd[2 * i], a1 = rowsd[2 * i + 1];
        for (int j = 0; j < n; ++j) {
            const uint4 t0 = rowsd[2 * j], t1 = rowsd[2 * j + 1];
            const int d = __popc(a0.x ^ t0.x) + __popc(a0.y ^ t0.y) + __popc(a0.z ^ t0.z) + __popc(a0.w ^ t0.w) +
                          __popc(a1.x ^ t1.x) + __popc(a1.y ^ t1.y) + __popc(a1.z ^ t1.z) + __popc(a1.w ^ t1.w);
            D[i * n + j] = (unsigned short)d;
        }
    }
    const int k = (int)(0.5 * (n - 1));  // index of the median in the sorted row
    unsigned bestkey = 0xFFFFFFFFu;       // (median << 16) | row
    for (int i = lane; i < n; i += 64) {
        int lo = 0, hi = 256;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            int c = 0;
            for (int j = 0; j < n; ++j) c += D[i * n + j] <= mid;
            if (c >= k + 1) hi = mid; else lo = mid + 1;
        }
        bestkey = min(bestkey, ((unsigned)lo << 16) | (unsigned)i);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) bestkey = min(bestkey, (unsigned)__shfl_xor((int)bestkey, d, 64));
    if (lane == 0) { best_idx[p] = (int)(bestkey & 0xFFFF); best_median[p] = (int)(bestkey >> 16); }
}

// the matrix-core kernel carries the train index in 13 bits; larger train sets (and SLAMIT_HAMMING_VALU=1, for A/B
// measurements) take the xor / popcount kernel.  Both give the same (index, best, second).
static bool hm_use_mfma(int max_train) {
    static const bool off = getenv("SLAMIT_HAMMING_VALU") != nullptr;
    return !off && max_train <= HMM_MAX_TRAIN;
}

extern "C" {

int slamit_distinctive_batch(const uint8_t* desc, const int32_t* offsets, int npoints, int32_t* best_idx,
                             int32_t* best_median) {
    if (npoints < 0 || (npoints && (!offsets || !best_idx || !best_median))) return slamit_fail(SLAMIT_ERR_ARG, "slamit_distinctive_batch: bad argument");
    if (npoints == 0) return SLAMIT_OK;
    for (int p = 0; p < npoints; ++p) {
        const int n = offsets[p + 1] - offsets[p];
        if (n < 0 || offsets[0] < 0) return slamit_fail(SLAMIT_ERR_ARG, "slamit_distinctive_batch: offsets must be non-decreasing");
        if (n > SLAMIT_DISTINCTIVE_MAX) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_distinctive_batch: more than SLAMIT_DISTINCTIVE_MAX rows for one point");
    }
    const int total = offsets[npoints];
    if (total && !desc) return slamit_fail(SLAMIT_ERR_ARG, "slamit_distinctive_batch: null descriptors");
    // one pinned staging block + one device slab per host thread (slamit_internal.h): [descriptors | offsets | out], one copy each way
    const size_t o_off = ((size_t)total * 32 + 255) & ~(size_t)255, o_out = o_off + ((sizeof(int) * ((size_t)npoints + 1) + 255) & ~(size_t)255);
    const size_t bytes = o_out + sizeof(int) * 2 * (size_t)npoints;
    const int cur_dev = slamit_default_device();   // slamit_set_device() of this thread, else the current device
    SLAMIT_USE_DEVICE(cur_dev);
    static thread_local SlamitScratch S;
    hipError_t e = slamit_scratch_reserve(S, cur_dev, bytes);
    if (e == hipSuccess) {
        if (total) memcpy(S.host, desc, (size_t)total * 32);
        memcpy(S.host + o_off, offsets, sizeof(int) * ((size_t)npoints + 1));
        e = hipMemcpyAsync(S.dev, S.host, o_out, hipMemcpyHostToDevice, S.st);
    }
    if (e == hipSuccess) {
        int* dout = reinterpret_cast<int*>(S.dev + o_out);
        hipLaunchKernelGGL(distinctive_kernel, dim3(npoints), dim3(64), 0, S.st, S.dev, reinterpret_cast<const int*>(S.dev + o_off), dout, dout + npoints);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(S.host + o_out, S.dev + o_out, bytes - o_out, hipMemcpyDeviceToHost, S.st);
    if (e == hipSuccess) e = hipStreamSynchronize(S.st);
    if (e == hipSuccess) {
        memcpy(best_idx, S.host + o_out, sizeof(int) * npoints);
        memcpy(best_median, S.host + o_out + sizeof(int) * npoints, sizeof(int) * npoints);
    }
    if (e != hipSuccess) return slamit_fail_hip(e, "slamit_distinctive_batch");
    return SLAMIT_OK;
}

int slamit_hamming_best2_batch_dev(const uint8_t* d_q, const int32_t* d_nq, size_t q_stride, const uint8_t* d_t,
                                   const int32_t* d_nt, size_t t_stride, int npairs, int max_n,
                                   int32_t* d_best_idx, int32_t* d_best, int32_t* d_second, size_t out_stride,
                                   int device, void* stream) {
    if (npairs < 0 || max_n < 0 || !d_q || !d_t || !d_nq || !d_nt || !d_best_idx || !d_best || !d_second)
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_hamming_best2_batch_dev: bad argument");
    if ((q_stride & 15) || (t_stride & 15) || ((uintptr_t)d_q & 15) || ((uintptr_t)d_t & 15))
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_hamming_best2_batch_dev: descriptors must be 16-byte aligned");
    if (npairs == 0 || max_n == 0) return SLAMIT_OK;
    if (max_n > SLAMIT_HAMMING_MAX_TRAIN) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_hamming_best2_batch_dev: more than SLAMIT_HAMMING_MAX_TRAIN descriptors per set");
    SLAMIT_USE_DEVICE(device);
    if (hm_use_mfma(max_n)) {
        hipLaunchKernelGGL(hamming_best2_mfma_kernel, dim3((max_n + 32 * HMM_NT - 1) / (32 * HMM_NT), npairs), dim3(64 * HMM_WAVES), 0, (hipStream_t)stream, d_q, d_nq, 0, q_stride,
                           d_t, d_nt, 0, t_stride, d_best_idx, d_best, d_second, out_stride);
    } else {
        dim3 grid((max_n + HM_QPB - 1) / HM_QPB, npairs);
        hipLaunchKernelGGL(hamming_best2_kernel, grid, dim3(256), 0, (hipStream_t)stream, d_q, d_nq, 0, q_stride, d_t, d_nt,
                           0, t_stride, d_best_idx, d_best, d_second, out_stride);
    }
    HIP_TRY(hipGetLastError());
    return SLAMIT_OK;
}

int slamit_hamming_best2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* best_idx, int32_t* best,
                         int32_t* second) {
    if (nq < 0 || nt < 0 || (nq && (!q || !best_idx || !best || !second)) || (nt && !t))
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_hamming_best2: bad argument");
    if (nq == 0) return SLAMIT_OK;
    if (nt > SLAMIT_HAMMING_MAX_TRAIN) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_hamming_best2: more than SLAMIT_HAMMING_MAX_TRAIN train descriptors");
    // one pinned staging block + one device slab per host thread: [query | train | out], one copy each way
    const size_t o_t = ((size_t)nq * 32 + 255) & ~(size_t)255, o_out = o_t + ((std::max<size_t>((size_t)nt * 32, 32) + 255) & ~(size_t)255);
    const size_t bytes = o_out + sizeof(int) * 3 * (size_t)nq;
    const int cur_dev = slamit_default_device();   // slamit_set_device() of this thread, else the current device
    SLAMIT_USE_DEVICE(cur_dev);
    static thread_local SlamitScratch S;
    hipError_t e = slamit_scratch_reserve(S, cur_dev, bytes);
    if (e == hipSuccess) {
        memcpy(S.host, q, (size_t)nq * 32);
        if (nt) memcpy(S.host + o_t, t, (size_t)nt * 32);
        e = hipMemcpyAsync(S.dev, S.host, o_out, hipMemcpyHostToDevice, S.st);
    }
    if (e == hipSuccess) {
        int* dout = reinterpret_cast<int*>(S.dev + o_out);
        if (hm_use_mfma(nt))
            hipLaunchKernelGGL(hamming_best2_mfma_kernel, dim3((nq + 32 * HMM_NT - 1) / (32 * HMM_NT), 1), dim3(64 * HMM_WAVES), 0, S.st, S.dev, (const int*)nullptr, nq,
                               (size_t)0, S.dev + o_t, (const int*)nullptr, nt, (size_t)0, dout, dout + nq, dout + 2 * (size_t)nq, (size_t)0);
        else
            hipLaunchKernelGGL(hamming_best2_kernel, dim3((nq + HM_QPB - 1) / HM_QPB, 1), dim3(256), 0, S.st, S.dev, (const int*)nullptr, nq,
                               (size_t)0, S.dev + o_t, (const int*)nullptr, nt, (size_t)0, dout, dout + nq, dout + 2 * (size_t)nq, (size_t)0);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(S.host + o_out, S.dev + o_out, bytes - o_out, hipMemcpyDeviceToHost, S.st);
    if (e == hipSuccess) e = hipStreamSynchronize(S.st);
    if (e == hipSuccess) {
        const int* o = reinterpret_cast<const int*>(S.host + o_out);
        memcpy(best_idx, o, sizeof(int) * nq); memcpy(best, o + nq, sizeof(int) * nq); memcpy(second, o + 2 * (size_t)nq, sizeof(int) * nq);
    }
    if (e != hipSuccess) return slamit_fail_hip(e, "slamit_hamming_best2");
    return SLAMIT_OK;
}

int slamit_hamming_matrix(const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* out) {
    if (nq < 0 || nt < 0 || ((nq && nt) && (!q || !t || !out))) return slamit_fail(SLAMIT_ERR_ARG, "slamit_hamming_matrix: bad argument");
    if (nq == 0 || nt == 0) return SLAMIT_OK;
    const int cur_dev = slamit_default_device();
    SLAMIT_USE_DEVICE(cur_dev);
    // one pinned staging block + one device slab per host thread: [query | train | matrix], one copy each way
    const size_t o_t = ((size_t)nq * 32 + 255) & ~(size_t)255, o_out = o_t + (((size_t)nt * 32 + 255) & ~(size_t)255);
    const size_t bytes = o_out + sizeof(uint16_t) * (size_t)nq * nt;
    static thread_local SlamitScratch S;
    hipError_t e = slamit_scratch_reserve(S, cur_dev, bytes);
    if (e == hipSuccess) {
        memcpy(S.host, q, (size_t)nq * 32);
        memcpy(S.host + o_t, t, (size_t)nt * 32);
        e = hipMemcpyAsync(S.dev, S.host, o_out, hipMemcpyHostToDevice, S.st);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(hamming_matrix_kernel, dim3((nt + 63) / 64, (nq + 63) / 64), dim3(256), 0, S.st, S.dev, nq, S.dev + o_t, nt,
                           reinterpret_cast<uint16_t*>(S.dev + o_out));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(S.host + o_out, S.dev + o_out, bytes - o_out, hipMemcpyDeviceToHost, S.st);
    if (e == hipSuccess) e = hipStreamSynchronize(S.st);
    if (e == hipSuccess) memcpy(out, S.host + o_out, bytes - o_out);
    if (e != hipSuccess) return slamit_fail_hip(e, "slamit_hamming_matrix");
    return SLAMIT_OK;
}

}  // extern "C"
