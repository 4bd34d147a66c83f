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

#include <algorithm>
#include <vector>

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
// K1 (per level, the path used when planes are 4-byte aligned): one lane = 4 adjacent dst pixels of
// TWO adjacent dst rows, items flattened over (row pair, group) so every lane of every wave is busy.  All
// per-column and per-row arithmetic the reference does at run time is in two host-built tables
// (orbk_resize_tables):
//   column group (3 x uint4): byte offset of the 12-byte source window | byte shift | offsets of its 2nd and
//       3rd dword (clamped into the row) ; 4 x v_perm selector that lifts (left tap, right tap) of one dst
//       column out of the 8-byte window as two zero-extended halfwords ; 4 x (ialpha0 | ialpha1 << 16)
//   row (uint2): sy0 | sy1 << 16 (clamped) ; ibeta0 | ibeta1 << 16
// A source row costs 3 aligned dword loads, 2 v_alignbyte, 4 v_perm and 4 v_dot2_u32_u16 (the horizontal pass
// of one dst pixel is ONE dot product); no LDS, no halo, no barrier.
// --------------------------------------------------------------------------------------------
typedef unsigned short rs_us2 __attribute__((ext_vector_type(2)));
typedef unsigned rs_u3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ uint32_t rs_dot2(uint32_t taps, uint32_t alpha) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(rs_us2, taps), __builtin_bit_cast(rs_us2, alpha), 0u, false);
}

template <int RP>   // RP row pairs (2 * RP dst rows) per lane; S / D: the frame's source and destination planes
__device__ __forceinline__ void rs_item(const int item, const uint8_t* S, const unsigned ss, const unsigned sbytes, uint8_t* D, const unsigned dstride,
                                        const uint4* __restrict__ coltab, const uint2* __restrict__ rowtab, const int ngroups,
                                        const unsigned inv_groups, const int dh) {
    const int yq = (int)__umulhi((unsigned)item, inv_groups);
    const int g = item - yq * ngroups;
    // uniform 64-bit bases + 32-bit per-lane offsets (planes are far below 4 GB): global_load saddr + voffset
    const uint8_t* ct = reinterpret_cast<const uint8_t*>(coltab) + 48u * (unsigned)g;
    const uint4 c0 = *reinterpret_cast<const uint4*>(ct), c1 = *reinterpret_cast<const uint4*>(ct + 16u);
    const uint32_t c2x = *reinterpret_cast<const uint32_t*>(ct + 32u);
    // rows past the bottom are clamped to the last row: they are produced again (same bytes), never skipped, so the
    // whole lane is straight-line code with every load issued before the first use
    int yrow[2 * RP];
    uint2 rt[2 * RP];
    {   // the table carries copies of its last row behind it: the lane's rows are one run of 16-byte pairs (RP loads, not 2 RP)
        const uint4* rp4 = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(rowtab) + 16u * (unsigned)(RP * yq));
#pragma unroll
        for (int r = 0; r < RP; ++r) {
            const uint4 v = rp4[r];
            rt[2 * r] = make_uint2(v.x, v.y); rt[2 * r + 1] = make_uint2(v.z, v.w);
        }
#pragma unroll
        for (int r = 0; r < 2 * RP; ++r) yrow[r] = min(2 * RP * yq + r, dh - 1);
    }
    const unsigned b = c0.x & 0xFFFFu, sh = (c0.x >> 16) & 3u, off1 = (c0.x >> 20) & 15u, off2 = (c0.x >> 24) & 15u;
    const uint32_t sel[4] = {c0.y, c0.z, c0.w, c1.x}, al[4] = {c1.y, c1.z, c1.w, c2x};
    uint32_t w[4 * RP][3];   // the 12-byte windows of the source rows (two per dst row)
    // ONE 12-byte buffer load per window instead of three dword loads (the kernel waits on vector-memory ISSUE: 72 % of its
    // wave cycles): the buffer resource bounds the plane, so the window of a row's last group may run past the row end
    // (into the next row, or past the plane's last byte, where the hardware returns 0) -- bytes no selector ever picks.
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(S), 0, sbytes, 0x00020000);
    (void)off1; (void)off2;
#pragma unroll
    for (int r = 0; r < 4 * RP; ++r) {
        const unsigned srow = (r & 1) ? (rt[r >> 1].x >> 16) : (rt[r >> 1].x & 0xFFFFu);
        const unsigned ro = __umul24(srow, ss) + b;   // rows < 2^16, pitch < 2^24
        const rs_u3 v = __builtin_amdgcn_raw_buffer_load_b96(rsrc, (int)ro, 0, 0);
        w[r][0] = v.x; w[r][1] = v.y; w[r][2] = v.z;
    }
#pragma unroll
    for (int k = 0; k < 2 * RP; ++k) {
        uint32_t h[2][4];   // horizontal pass: h[source row][column] = (left * ialpha0 + right * ialpha1) >> 4
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t lo = __builtin_amdgcn_alignbyte(w[2 * k + r][1], w[2 * k + r][0], sh), hi = __builtin_amdgcn_alignbyte(w[2 * k + r][2], w[2 * k + r][1], sh);
#pragma unroll
            for (int j = 0; j < 4; ++j) h[r][j] = rs_dot2(__builtin_amdgcn_perm(hi, lo, sel[j]), al[j]) >> 4;
        }
        const unsigned b0 = rt[k].y & 0xFFFFu, b1 = rt[k].y >> 16;
        uint32_t out = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // (b * (h >> 4)) >> 16 with b <= 2048, h >> 4 <= 32640: 24-bit multiplies
            const uint32_t v = ((__umul24(b0, h[0][j]) >> 16) + (__umul24(b1, h[1][j]) >> 16) + 2u) >> 2;
            out |= (v & 0xFFu) << (8 * j);
        }
        *reinterpret_cast<uint32_t*>(D + (__umul24((unsigned)yrow[k], dstride) + 4u * (unsigned)g)) = out;
    }
}

template <int RP>
__global__ __launch_bounds__(256) void resize_rows4_kernel(
    const uint8_t* __restrict__ src, size_t sstride, size_t sframe,
    uint8_t* __restrict__ dst, size_t dstride, size_t dframe,
    const uint4* __restrict__ coltab, const uint2* __restrict__ rowtab, int ngroups, unsigned inv_groups, int nitems, int dh, unsigned sbytes) {
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= nitems) return;
    rs_item<RP>(item, src + (size_t)blockIdx.y * sframe, (unsigned)sstride, sbytes, dst + (size_t)blockIdx.y * dframe, (unsigned)dstride, coltab, rowtab,
                ngroups, inv_groups, dh);
}

// The same pass with EIGHT adjacent dst pixels per lane and row: at the pyramid's scale factors (<= ~1.3) their sixteen taps lie
// inside one 16-byte source window, so a source row costs ONE 16-byte buffer load per eight pixels instead of one 12-byte
// load per four, and a dst row one 8-byte store -- the kernel waits on vector-memory ISSUE, not on bytes.  After the
// per-lane byte shift the taps of pixels 0 .. 3 lie in dwords (a0, a1) and those of pixels 4 .. 7 in (a1, a2): static
// register pairs (orbk_resize_tables8 checks it and refuses other geometries, which keep the four-pixel kernel).
// Column group (5 x uint4): byte offset of the window | shift << 16 ; 8 x v_perm selector ; 8 x (ialpha0 | ialpha1 << 16).
typedef unsigned rs_u4 __attribute__((ext_vector_type(4)));
template <int RP>
__device__ __forceinline__ void rs_item8(const int item, const uint8_t* S, const unsigned ss, const unsigned sbytes, uint8_t* D, const unsigned dstride,
                                         const uint4* __restrict__ coltab, const uint2* __restrict__ rowtab, const int ngroups,
                                         const unsigned inv_groups, const int dh) {
    const int yq = (int)__umulhi((unsigned)item, inv_groups);
    const int g = item - yq * ngroups;
    const uint8_t* ct = reinterpret_cast<const uint8_t*>(coltab) + 80u * (unsigned)g;
    const uint4 c0 = *reinterpret_cast<const uint4*>(ct), c1 = *reinterpret_cast<const uint4*>(ct + 16u), c2 = *reinterpret_cast<const uint4*>(ct + 32u),
                c3 = *reinterpret_cast<const uint4*>(ct + 48u);
    const uint32_t c4x = *reinterpret_cast<const uint32_t*>(ct + 64u);
    int yrow[2 * RP];
    uint2 rt[2 * RP];
    {
        const uint4* rp4 = reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(rowtab) + 16u * (unsigned)(RP * yq));
#pragma unroll
        for (int r = 0; r < RP; ++r) {
            const uint4 v = rp4[r];
            rt[2 * r] = make_uint2(v.x, v.y); rt[2 * r + 1] = make_uint2(v.z, v.w);
        }
#pragma unroll
        for (int r = 0; r < 2 * RP; ++r) yrow[r] = min(2 * RP * yq + r, dh - 1);
    }
    const unsigned b = c0.x & 0xFFFFu, sh = (c0.x >> 16) & 3u;
    const uint32_t sel[8] = {c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x}, al[8] = {c2.y, c2.z, c2.w, c3.x, c3.y, c3.z, c3.w, c4x};
    uint32_t w[4 * RP][4];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(S), 0, sbytes, 0x00020000);
#pragma unroll
    for (int r = 0; r < 4 * RP; ++r) {
        const unsigned srow = (r & 1) ? (rt[r >> 1].x >> 16) : (rt[r >> 1].x & 0xFFFFu);
        const unsigned ro = __umul24(srow, ss) + b;
        const rs_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)ro, 0, 0);
        w[r][0] = v.x; w[r][1] = v.y; w[r][2] = v.z; w[r][3] = v.w;
    }
#pragma unroll
    for (int k = 0; k < 2 * RP; ++k) {
        uint32_t h[2][8];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const uint32_t* ww = w[2 * k + r];
            const uint32_t a0 = __builtin_amdgcn_alignbyte(ww[1], ww[0], sh), a1 = __builtin_amdgcn_alignbyte(ww[2], ww[1], sh), a2 = __builtin_amdgcn_alignbyte(ww[3], ww[2], sh);
#pragma unroll
            for (int j = 0; j < 4; ++j) h[r][j] = rs_dot2(__builtin_amdgcn_perm(a1, a0, sel[j]), al[j]) >> 4;
#pragma unroll
            for (int j = 4; j < 8; ++j) h[r][j] = rs_dot2(__builtin_amdgcn_perm(a2, a1, sel[j]), al[j]) >> 4;
        }
        const unsigned b0 = rt[k].y & 0xFFFFu, b1 = rt[k].y >> 16;
        uint32_t out[2] = {0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t v = ((__umul24(b0, h[0][j]) >> 16) + (__umul24(b1, h[1][j]) >> 16) + 2u) >> 2;
            out[j >> 2] |= (v & 0xFFu) << (8 * (j & 3));
        }
        *reinterpret_cast<uint2*>(D + (__umul24((unsigned)yrow[k], dstride) + 8u * (unsigned)g)) = make_uint2(out[0], out[1]);
    }
}

template <int RP>
__global__ __launch_bounds__(256) void resize_rows8_kernel(
    const uint8_t* __restrict__ src, size_t sstride, size_t sframe,
    uint8_t* __restrict__ dst, size_t dstride, size_t dframe,
    const uint4* __restrict__ coltab, const uint2* __restrict__ rowtab, int ngroups, unsigned inv_groups, int nitems, int dh, unsigned sbytes) {
    const int item = blockIdx.x * 256 + threadIdx.x;
    if (item >= nitems) return;
    rs_item8<RP>(item, src + (size_t)blockIdx.y * sframe, (unsigned)sstride, sbytes, dst + (size_t)blockIdx.y * dframe, (unsigned)dstride, coltab, rowtab,
                 ngroups, inv_groups, dh);
}

// Host: tables of resize_rows4_kernel for one level from the reference-shaped xofs/ialpha/yofs/ibeta tables.
// Coefficients are non-negative (bilinear) and <= 2048, rows/columns < 65536 (checked by the caller).
// Returns false when the geometry does not fit the kernel (scale factor > 2.3: taps of one group further than
// 8 bytes apart; planes of 64K pixels or more): the caller then uses the fused / generic kernels.
bool orbk_resize_tables(int dw, int dh, int sw, int sh, const int* xofs, const short* ialpha, const int* yofs,
                        const short* ibeta, std::vector<uint32_t>& col, std::vector<uint32_t>& row) {
    const int ng = (dw + 3) / 4, row_end = (sw + 3) & ~3;
    if (sw >= 65536 || sh >= 65536) return false;
    col.assign((size_t)ng * 12, 0u);
    for (int g = 0; g < ng; ++g) {
        int L[4], R[4];
        uint32_t a[4];
        for (int j = 0; j < 4; ++j) {
            const int x = 4 * g + j;
            if (x < dw) {
                L[j] = xofs[x]; R[j] = std::min(xofs[x] + 1, sw - 1);
                a[j] = (uint32_t)(unsigned short)ialpha[2 * x] | ((uint32_t)(unsigned short)ialpha[2 * x + 1] << 16);
            } else { L[j] = L[0]; R[j] = L[0]; a[j] = 0; }   // padding pixels of the last dword: written as 0
        }
        const int base = L[0] & ~3, s = L[0] & 3;
        const int off1 = base + 8 <= row_end ? 4 : 0, off2 = base + 12 <= row_end ? 8 : off1;
        uint32_t* c = &col[(size_t)g * 12];
        c[0] = (uint32_t)base | ((uint32_t)s << 16) | ((uint32_t)off1 << 20) | ((uint32_t)off2 << 24);
        for (int j = 0; j < 4; ++j) {   // byte index inside the 8 bytes that start at L[0]; selector byte 0x0C reads as zero
            if (L[j] < L[0] || R[j] < L[0] || L[j] - L[0] > 7 || R[j] - L[0] > 7 || ialpha[0] < 0) return false;
            c[1 + j] = (uint32_t)(L[j] - L[0]) | 0x0C00u | ((uint32_t)(R[j] - L[0]) << 16) | 0x0C000000u;
            c[5 + j] = a[j];
        }
    }
    row.assign((size_t)(dh + 7) * 2, 0u);   // + 7 copies of the last row: a lane loads its (up to 8) rows as whole 16-byte pairs
    for (int y = 0; y < dh; ++y) {
        const int sy0 = std::min(std::max(yofs[y], 0), sh - 1), sy1 = std::min(std::max(yofs[y] + 1, 0), sh - 1);
        row[2 * (size_t)y] = (uint32_t)sy0 | ((uint32_t)sy1 << 16);
        row[2 * (size_t)y + 1] = (uint32_t)(unsigned short)ibeta[2 * y] | ((uint32_t)(unsigned short)ibeta[2 * y + 1] << 16);
        if (ibeta[2 * y] < 0 || ibeta[2 * y + 1] < 0) return false;
    }
    for (int y = dh; y < dh + 7; ++y) { row[2 * (size_t)y] = row[2 * (size_t)(dh - 1)]; row[2 * (size_t)y + 1] = row[2 * (size_t)(dh - 1) + 1]; }
    for (int x = 0; x < dw; ++x) if (ialpha[2 * x] < 0 || ialpha[2 * x + 1] < 0) return false;
    return true;
}

// Column table of resize_rows8_kernel (the row table is orbk_resize_tables'); false when some group's taps do not lie as
// the kernel assumes -- pixels 0 .. 3 within bytes 0 .. 7 of the window that starts at the group's first left tap, pixels
// 4 .. 7 within bytes 4 .. 11 -- or the 8-byte stores of the last group would pass the row pitch.
bool orbk_resize_tables8(int dw, int sw, size_t dstride, const int* xofs, const short* ialpha, std::vector<uint32_t>& col) {
    const int ng = (dw + 7) / 8;
    if (sw >= 65536 || (size_t)ng * 8 > dstride) return false;
    col.assign((size_t)ng * 20, 0u);
    for (int g = 0; g < ng; ++g) {
        int L[8], R[8];
        uint32_t a[8];
        const int L0 = xofs[8 * g];
        for (int j = 0; j < 8; ++j) {
            const int x = 8 * g + j;
            if (x < dw) {
                L[j] = xofs[x]; R[j] = std::min(xofs[x] + 1, sw - 1);
                if (ialpha[2 * x] < 0 || ialpha[2 * x + 1] < 0) return false;
                a[j] = (uint32_t)(unsigned short)ialpha[2 * x] | ((uint32_t)(unsigned short)ialpha[2 * x + 1] << 16);
            } else { L[j] = R[j] = L0 + (j >= 4 ? 4 : 0); a[j] = 0; }   // padding pixels of the last group: written as 0
        }
        uint32_t* c = &col[(size_t)g * 20];
        c[0] = (uint32_t)(L0 & ~3) | ((uint32_t)(L0 & 3) << 16);
        for (int j = 0; j < 8; ++j) {
            const int lo = j >= 4 ? 4 : 0;   // byte index inside the dword pair the pixel reads
            const int bl = L[j] - L0 - lo, br = R[j] - L0 - lo;
            if (bl < 0 || bl > 7 || br < 0 || br > 7) return false;
            c[1 + j] = (uint32_t)bl | 0x0C00u | ((uint32_t)br << 16) | 0x0C000000u;
            c[9 + j] = a[j];
        }
    }
    return true;
}

void orbk_resize_rows8(hipStream_t st, const uint8_t* src, size_t sstride, size_t sframe, int sh, uint8_t* dst, int dw, int dh,
                       size_t dstride, size_t dframe, const uint32_t* d_col8, const uint32_t* d_row, int nframes) {
    static const int rp_env = getenv("SLAMIT_RESIZE8_RP") ? atoi(getenv("SLAMIT_RESIZE8_RP")) : 0;
    const int rp = rp_env == 1 || rp_env == 2 || rp_env == 4 ? rp_env : 2;
    const int ng = (dw + 7) / 8, nitems = ng * ((dh + 2 * rp - 1) / (2 * rp));
    const unsigned inv = (unsigned)((0x100000000ull + (unsigned)ng - 1) / (unsigned)ng);
    const dim3 grid((nitems + 255) / 256, nframes);
#define RS_LAUNCH8(RP) hipLaunchKernelGGL(resize_rows8_kernel<RP>, grid, dim3(256), 0, st, src, sstride, sframe, dst, dstride, dframe, \
                                           reinterpret_cast<const uint4*>(d_col8), reinterpret_cast<const uint2*>(d_row), ng, inv, nitems, dh, (unsigned)(sstride * (size_t)sh))
    if (rp == 1) RS_LAUNCH8(1); else if (rp == 2) RS_LAUNCH8(2); else RS_LAUNCH8(4);
#undef RS_LAUNCH8
}

void orbk_resize_rows4(hipStream_t st, const uint8_t* src, size_t sstride, size_t sframe, int sh, uint8_t* dst, int dw, int dh,
                       size_t dstride, size_t dframe, const uint32_t* d_col, const uint32_t* d_row, int nframes) {
    static const int rp_env = getenv("SLAMIT_RESIZE_RP") ? atoi(getenv("SLAMIT_RESIZE_RP")) : 0;
    const int rp = rp_env == 1 || rp_env == 2 || rp_env == 4 ? rp_env : 2;
    const int ng = (dw + 3) / 4, nitems = ng * ((dh + 2 * rp - 1) / (2 * rp));
    const unsigned inv = (unsigned)((0x100000000ull + (unsigned)ng - 1) / (unsigned)ng);
    const dim3 grid((nitems + 255) / 256, nframes);
#define RS_LAUNCH(RP) hipLaunchKernelGGL(resize_rows4_kernel<RP>, grid, dim3(256), 0, st, src, sstride, sframe, dst, dstride, dframe, \
                                          reinterpret_cast<const uint4*>(d_col), reinterpret_cast<const uint2*>(d_row), ng, inv, nitems, dh, (unsigned)(sstride * (size_t)sh))
    if (rp == 1) RS_LAUNCH(1); else if (rp == 2) RS_LAUNCH(2); else RS_LAUNCH(4);
#undef RS_LAUNCH
}

// --------------------------------------------------------------------------------------------
// K1 (fused): the whole pyramid in ONE launch.  Workgroup (region, frame) loads its level-0 patch
// once, then produces level 1, 2, ... each from the previous level kept in LDS (two ping-pong
// buffers), storing only the pixels it owns.  The chain of integer roundings is exactly the
// reference's (level l is always computed from level l-1), but no level is ever re-read from HBM
// and six dependent launches disappear.  Boxes come from the host (orb_api.hip: build_pyr_boxes).
// --------------------------------------------------------------------------------------------
#define PYR_THREADS 512   // upper bound; the launch picks 256 or 512

__global__ __launch_bounds__(PYR_THREADS) void pyramid_fused_kernel(
    const OrbLevel* __restrict__ levels, int nlevels, const PyrBox* __restrict__ boxes,
    const PyrTabs* __restrict__ tabs, const uint8_t* __restrict__ img0, size_t img0_stride, size_t img0_frame,
    uint8_t* __restrict__ pyr, int bufA_bytes, int l_first, int l_last) {
    extern __shared__ __attribute__((aligned(16))) uint8_t pbuf[];
    uint8_t* buf[2] = {pbuf, pbuf + bufA_bytes};
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int PYR_ROWS = blockDim.x >> 6, NT = blockDim.x;
    const int frame = blockIdx.y;
    const PyrBox* B = boxes + (size_t)blockIdx.x * nlevels;
    // source patch of the segment's first level (its x0 is a multiple of 4: whole dwords when the plane is
    // 4-byte aligned — pyramid planes always are, the caller's level 0 usually)
    {
        const PyrBox b = B[l_first];
        const int nw = b.nx1 - b.nx0, pitch = (nw + 3) & ~3;
        const uint8_t* S;
        size_t sstride;
        if (l_first == 0) { S = img0 + (size_t)frame * img0_frame; sstride = img0_stride; }
        else { S = pyr + levels[l_first].plane_off + (size_t)frame * levels[l_first].plane_bytes; sstride = (size_t)levels[l_first].stride; }
        const bool aligned = ((((uintptr_t)S) | sstride) & 3) == 0;
        const int w0 = levels[l_first].w;
        if (aligned) {
            const int nd = pitch >> 2;
            for (int y = b.ny0 + ty; y < b.ny1; y += PYR_ROWS)
                for (int d = tx; d < nd; d += 64) {
                    const int x = b.nx0 + 4 * d;
                    uint32_t v;
                    if (x + 4 <= w0) v = *reinterpret_cast<const uint32_t*>(S + (size_t)y * sstride + x);
                    else { v = 0; for (int j = 0; j < 4; ++j) if (x + j < w0) v |= (uint32_t)S[(size_t)y * sstride + x + j] << (8 * j); }
                    *reinterpret_cast<uint32_t*>(&buf[0][(y - b.ny0) * pitch + 4 * d]) = v;
                }
        } else {
            for (int y = b.ny0 + ty; y < b.ny1; y += PYR_ROWS)
                for (int x = b.nx0 + tx; x < b.nx1; x += 64)
                    buf[0][(y - b.ny0) * pitch + (x - b.nx0)] = S[(size_t)y * sstride + x];
        }
    }
    __syncthreads();
    for (int l = l_first + 1; l <= l_last; ++l) {
        const PyrBox b = B[l], pb = B[l - 1];
        const OrbLevel& L = levels[l];
        const int sw = levels[l - 1].w, sh = levels[l - 1].h;
        const PyrTabs T = tabs[l];
        const uint8_t* src = buf[(l - 1 - l_first) & 1];
        uint8_t* dst = buf[(l - l_first) & 1];
        const int spitch = ((pb.nx1 - pb.nx0) + 3) & ~3, dpitch = ((b.nx1 - b.nx0) + 3) & ~3;
        uint8_t* plane = pyr + L.plane_off + (size_t)frame * L.plane_bytes;
        // a lane keeps the coefficients of its (<= 4) columns in registers; rows are wave-uniform, so
        // their tables come through the scalar unit
        const int nw = b.nx1 - b.nx0;
        int cx0[4], cx1[4], ca0[4], ca1[4];
        unsigned ownmask = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = b.nx0 + tx + 64 * j;
            if (tx + 64 * j < nw) {
                const int sx = T.xofs[x];
                cx0[j] = sx - pb.nx0;
                cx1[j] = min(sx + 1, sw - 1) - pb.nx0;
                ca0[j] = T.ialpha[2 * x];
                ca1[j] = T.ialpha[2 * x + 1];
                if (x >= b.ox0 && x < b.ox1) ownmask |= 1u << j;
            } else { cx0[j] = cx1[j] = 0; ca0[j] = ca1[j] = 0; }
        }
        // the row tables of this level's rows go through LDS once (a dependent scalar load per row
        // would stall every iteration)
        __shared__ int s_sy0[256], s_sy1[256], s_b01[256];
        const int nh = b.ny1 - b.ny0;
        for (int i = threadIdx.x; i < nh; i += NT) {
            const int y = b.ny0 + i, sy = T.yofs[y];
            s_sy0[i] = min(max(sy, 0), sh - 1) - pb.ny0;
            s_sy1[i] = min(max(sy + 1, 0), sh - 1) - pb.ny0;
            s_b01[i] = (int)(unsigned short)T.ibeta[2 * y] | ((int)T.ibeta[2 * y + 1] << 16);
        }
        __syncthreads();
        const int wy = __builtin_amdgcn_readfirstlane(ty);
        for (int y = b.ny0 + wy; y < b.ny1; y += PYR_ROWS) {
            const int sy0 = s_sy0[y - b.ny0], sy1 = s_sy1[y - b.ny0];
            const int b01 = s_b01[y - b.ny0];
            const int b0 = (short)(b01 & 0xFFFF), b1 = b01 >> 16;
            const uint8_t* S0 = src + sy0 * spitch;
            const uint8_t* S1 = src + sy1 * spitch;
            const bool own_y = y >= b.oy0 && y < b.oy1;
            uint8_t* drow = dst + (y - b.ny0) * dpitch + tx;
            uint8_t* prow = plane + (size_t)y * L.stride + b.nx0 + tx;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (tx + 64 * j < nw) {
                    const int r0 = S0[cx0[j]] * ca0[j] + S0[cx1[j]] * ca1[j];
                    const int r1 = S1[cx0[j]] * ca0[j] + S1[cx1[j]] * ca1[j];
                    const int v = ((((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2) & 0xFF;
                    drow[64 * j] = (uint8_t)v;
                    if (own_y && ((ownmask >> j) & 1)) prow[64 * j] = (uint8_t)v;
                }
            }
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------------------------
// K2: FAST-9/16 per cell (ORBextractor.cc:805-849; cv::FAST_t<16>, cornerScore<16>, NMS).  ONE WAVEFRONT = one cell
// of one level of one frame, four cells per workgroup, no workgroup barrier.
//
// The cell's window (<= 65x65 bytes) is staged in the wave's LDS slice once.  Per attempt (iniThFAST, then minThFAST
// if the cell kept nothing, ORBextractor.cc:827-833):
//   A. compass pre-test, byte-parallel on 4 pixels per lane: a 9-arc of the 16-ring always holds two ADJACENT compass
//      points, so a corner needs two adjacent ones brighter than v + t, or two darker than v - t;
//   B. the survivors of a step (256 pixels) are appended to ONE entry stack (tile offset, which polarities passed) in raster
//      order: a lane counts its pixels, a wave prefix sum (DPP) gives its first slot, four predicated stores write them;
//   C. whenever 128 entries wait they are scored, TWO per lane on packed 16-bit lanes, with the exact
//      cornerScore: max over the sixteen 9-arcs of (min over the arc) - centre - 1.  A "darker" candidate is scored on
//      the complemented bytes (255 - v turns it into a "brighter" one with the same score), so both halves run the same
//      min / max network; a pixel that passed in both polarities is scored as darker and queued again as brighter;
//      scores >= t go to a byte tile (stored + 1) and the corner to a list;
//   D. NMS over the listed corners (strict >, 8 neighbours' stored scores; cv::FAST zero-fills outside the cell's scan
//      area and below t), kept entries compacted in place.
// A candidate leaves as one u64: (score << 32) | order, order = (cell << 12) | (y_local << 6) | x_local = its position in
// the reference's vToDistributeKeys (cell row, cell column, y, x); the octree only needs that order to break response
// ties, so the list itself is unordered.
//
// The kernel is bound by the vector instructions a SIMD issues (rocprofv3 SQ counters): every step
// is wave-synchronous (no s_barrier), appends use ballot + mbcnt with the running count in a scalar register (no LDS
// atomics), and a wave needs ~5 KB of LDS for the usual <= 40-pixel cells so 7 waves fit per SIMD.
// --------------------------------------------------------------------------------------------
#define SC_COL0 4   // scan pixel x sits at LDS column x + 4 in both the image tile and the score tile

__device__ __forceinline__ int imax3(int a, int b, int c) { return max(max(a, b), c); }

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

#define FAST_ENT_CAP 386   // pixel entries (u16): < 128 left over + <= 256 of a pre-test step (a pass pops 128 and re-queues at most as many); the last slot takes the stores of pixels that did not pass

// LDS bytes of one wave: image tile | score tile | pixel entries (u16) | corner / keypoint list (u16)
__host__ __device__ inline int fast_tile_bytes(int pitch, int tile_rows) { return (tile_rows * pitch + 16 + 15) & ~15; }
__host__ __device__ inline int fast_sc_bytes(int pitch, int sc_rows) { return (sc_rows * pitch + 15) & ~15; }
__host__ __device__ inline int fast_wave_bytes(int pitch, int tile_rows, int sc_rows, int kp_cap) {
    return (fast_tile_bytes(pitch, tile_rows) + fast_sc_bytes(pitch, sc_rows) + 2 * FAST_ENT_CAP + 2 * kp_cap + 15) & ~15;
}

// gfx950's three-input packed f16 min / max: two pixels per lane; the lanes hold the integers 0 .. 255 as f16 bit patterns
// (denormals, kept by the default FP mode) and are only compared
__device__ __forceinline__ uint32_t pk_min3_f16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t pk_max3_f16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t d;
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ uint32_t pk_max_f16(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// v + (this lane's bit of `mask`): one v_addc with the mask as carry-in
__device__ __forceinline__ unsigned add_flag(unsigned v, unsigned long long mask) {
    unsigned d;
    unsigned long long carry;
    asm("v_addc_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(d), "=s"(carry) : "v"(v), "s"(mask));
    return d;
}

// exclusive prefix sum of `v` over the wavefront: four DPP row shifts scan each 16-lane row, two row broadcasts carry the row
// totals on (lane 15 -> rows 1 and 3, lane 31 -> rows 2 and 3); `total` = the sum over all lanes (wave-uniform)
__device__ __forceinline__ int wave_exclusive_scan(int v, int& total) {
    int incl = v;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);   // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);   // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);   // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);   // row_shr:8
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1, 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2, 3
    total = __builtin_amdgcn_readlane(incl, 63);
    return incl - v;
}

#ifdef FAST_DIAG   // diagnostic builds only (tools/diag): per-phase wave cycles, one record per wave
#define FD_MAXW 65536
__device__ unsigned long long g_fast_ph[FD_MAXW * 8];
extern "C" int slamit_diag_fast(unsigned long long* out, int reset) {
    if (reset) { void* p; hipGetSymbolAddress(&p, HIP_SYMBOL(g_fast_ph)); return (int)hipMemset(p, 0, sizeof(unsigned long long) * FD_MAXW * 8); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fast_ph), sizeof(unsigned long long) * FD_MAXW * 8);
}
#define FD_DECL unsigned long long fd_ph[6] = {0, 0, 0, 0, 0, 0}, fd_prev = __builtin_amdgcn_s_memtime(), fd_surv = 0, fd_t0 = fd_prev
#define FD_STAMP(i) do { unsigned long long tn = __builtin_amdgcn_s_memtime(); fd_ph[i] += tn - fd_prev; fd_prev = tn; } while (0)
#define FD_FLUSH(npx) do { const int fd_w = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + wv; if (lane == 0 && fd_w < FD_MAXW) { \
    unsigned long long* o = g_fast_ph + 8 * (size_t)fd_w; for (int k = 0; k < 5; ++k) o[k] = fd_ph[k]; \
    o[5] = (fd_surv << 32) | (unsigned)(npx); o[6] = fd_t0; o[7] = fd_prev; } } while (0)
#else
#define FD_DECL
#define FD_STAMP(i)
#define FD_FLUSH(npx)
#endif

// The work of one wavefront on one cell.  `fsm` is the workgroup's dynamic LDS; the wave's slice is carved inside.
template <int PITCH>
__device__ __forceinline__ void fast_cell_wave(
    uint8_t* fsm, const int lane, const int wv, const int frame, const int cell,
    const FastTab& tab, const uint4* __restrict__ cells, int nlevels,
    const uint8_t* __restrict__ img0, unsigned img0_stride, size_t img0_frame,
    const uint8_t* __restrict__ pyr,
    unsigned long long* __restrict__ cand, size_t cand_frame_stride,
    int* __restrict__ cand_count, int iniTh, int minTh, int tile_rows, int sc_rows, int kp_cap) {
    FD_DECL;
    uint8_t* tile = fsm + wv * fast_wave_bytes(PITCH, tile_rows, sc_rows, kp_cap);
    uint8_t* sc = tile + fast_tile_bytes(PITCH, tile_rows);
    unsigned short* s_ent = reinterpret_cast<unsigned short*>(sc + fast_sc_bytes(PITCH, sc_rows));
    unsigned short* s_kp = s_ent + FAST_ENT_CAP;

    // the cell's geometry and the division it needs, precomputed on the host (orbk_fast_cells): one scalar
    // 16-byte load instead of ~400 instructions of level search and integer division per wave
    const uint4 ca = cells[cell];
    const int level = (int)(ca.x & 255u), c = (int)(ca.x >> 8);
    const int iniX = (int)(ca.y & 0xFFFFu), iniY = (int)(ca.y >> 16);
    const int cw = (int)(ca.z & 255u), ch = (int)((ca.z >> 8) & 255u);
    const unsigned inv_g = ca.w;
    const FastLevel& L = tab.lv[level];
    const int sw = cw - 6, sh = ch - 6;  // scan area = FAST's [3, n-3)

    const uint8_t* src;
    unsigned stride;
    if (level == 0) { src = img0 + (size_t)frame * img0_frame; stride = img0_stride; }
    else { src = pyr + L.plane_off + (size_t)frame * L.plane_bytes; stride = (unsigned)L.stride; }
    src += (size_t)iniY * stride + iniX;

    // ---- stage the window: window column j -> LDS column j + 1 (so that scan pixel x is at x + 4) ----
    {
        // LDS-DMA: a lane moves 16 bytes of a window row straight into the tile (global_load_lds_dwordx4: LDS address =
        // wave-uniform base + 16 * lane, so lane -> (row, 16-byte chunk) in tile order; the source address is per lane and
        // need not be aligned: tools/diag/ubench/glds_unaligned.hip).  No VGPR carries pixels, no v_alignbyte, no LDS store
        // instruction, and a caller's plane of any pitch and alignment takes the same path.  Only chunks that START inside the
        // window are fetched: the window ends >= 16 pixels before the end of the image row, so a chunk never leaves the row.
        static_assert(PITCH % 16 == 0, "a tile row is a whole number of 16-byte DMA chunks");
        constexpr int CPR = PITCH / 16;
        const int kact = (cw + 1 + 15) >> 4, nch = ch * CPR;
        for (int c0 = 0; c0 < nch; c0 += WAVE) {
            const int ci = c0 + lane;
            const int r = CPR == 3 ? (int)(__umul24(ci, 171) >> 9) : ci / CPR;   // ci < 256: floor(ci / 3)
            const int k = ci - r * CPR;
            if (ci < nch && k < kact)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src - 1 + __umul24(r, stride) + 16 * k),
                                                 (__attribute__((address_space(3))) void*)(tile + 16 * c0), 16, 0, 0);
        }
    }
    {   // zero the rows of the score tile this cell uses (plus the ring around them), 16 bytes per lane
        const int nz = ((sh + 2) * PITCH + 15) >> 4;
        for (int i = lane; i < nz; i += WAVE) reinterpret_cast<uint4*>(sc)[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA's LDS writes are ordered for this wave's reads by its own vmcnt wait
    wave_sync_lds();
    FD_STAMP(0);

    const int ngrp = (sw + 3) >> 2;          // groups of 4 scan pixels per row
    const int nitems = ngrp * sh;
    // pre-test mapping: lane -> (row within a step, column group).  A lane keeps its column group for the whole cell (rps
    // rows of ngrp groups per step; the 64 % ngrp lanes left over idle): column, tail mask and LDS address are loop invariants
    const int p_rsub = (int)(__umul24(lane, inv_g) >> 20), p_x0 = (lane - p_rsub * ngrp) * 4;
    const int rps = (int)(__umul24(WAVE, inv_g) >> 20);      // WAVE / ngrp rows per step (ngrp <= 16: cells are <= 64 px wide)
    const bool p_active = p_rsub < rps;
    unsigned p_vmask = 0x80808080u;                          // pixels of the row's last group beyond the scan row (0 .. 3) are masked
    if (p_x0 + 4 - sw > 0) p_vmask >>= 8 * (p_x0 + 4 - sw);
    const unsigned p_vmask6 = p_vmask >> 1;                  // a lane's flag word, per pixel byte: bit 7 = brighter passed, bit 6 = darker passed

    int th = iniTh;
    int nkp = 0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        // For bytes a (ring) and c (centre) let D = (a + ~c + r) >> 1 per byte (v_lerp_u8: 9-bit intermediate), r = parity of
        // T = t + 256.  a - c > t  <=>  a + (255 - c) >= T  <=>  D >= (T + 1) >> 1: a second lerp against 256 - that constant
        // leaves the answer in the byte's MSB.  c - a > t  <=>  a + (255 - c) <= 254 - t  ==>  D <= K2 = (254 - t + r) >> 1
        // (a necessary condition: at most one grey level looser, and the exact score decides): the MSB of a lerp of the SAME D
        // against 255 - K2 says "D > K2", i.e. NOT darker -- the inversion rides in the truth table of the v_bitop3 that
        // combines the compass points.
        const int T = th + 256;
        const unsigned rnd = (T & 1) ? 0x01010101u : 0u;
        const unsigned kcB = (unsigned)(256 - ((T + 1) >> 1)) * 0x01010101u;
        const unsigned kcD = (unsigned)(255 - ((254 - th + (T & 1)) >> 1)) * 0x01010101u;
        int nE = 0;                // pixel entries waiting
        int ncorner = 0;
        bool corner_overflow = false;

        // scoring: 128 entries (at the end of a cell: whatever is left) off the top of the entry stack, TWO pixels per lane on
        // packed 16-bit lanes.  Entry = (y * PITCH + x) << 2 | code, code 1 = darker, 2 = brighter, 3 = both (scored as darker now
        // and queued again as brighter); lanes without an entry score pixel (0, 0) and do not store.
        auto score_pass = [&](const int cnt) {
            wave_sync_lds();
            FD_STAMP(1);
#ifdef FAST_DIAG
            fd_surv += cnt;
#endif
            const int base = nE - cnt;
            const bool vlo = lane < cnt, vhi = lane + 64 < cnt;
            // lanes beyond the count all re-read the pass's first entry (one address: a broadcast, no bank conflict -- stale slots
            // would scatter their ring reads over the tile) and are kept out of every store below
            const unsigned elo = s_ent[base + (vlo ? lane : 0)], ehi = s_ent[base + (vhi ? lane + 64 : 0)];
            const unsigned long long vmlo = cnt >= 64 ? ~0ull : (1ull << cnt) - 1ull;
            const unsigned long long vmhi = cnt >= 128 ? ~0ull : cnt > 64 ? (1ull << (cnt - 64)) - 1ull : 0ull;
            // y * PITCH + x: top-left corner of the pixel's 7x7 neighbourhood in the tile, less SC_COL0 - 3
            const unsigned olo = elo >> 2, ohi = ehi >> 2;
            const uint8_t* qlo = tile + olo + (SC_COL0 - 3);
            const uint8_t* qhi = tile + ohi + (SC_COL0 - 3);
#ifdef FAST_DIAG_NO_XOR   // timing experiment only (wrong scores for "darker" entries)
            const uint32_t pm = 0u;
#else
            const uint32_t pm = __umul24(((ehi << 16) | elo) & 0x00010001u, 0xFFu);   // 0xFF in the half of an entry scored as "darker": complements its bytes
#endif
#define PX(dx, dy) ((((uint32_t)qhi[((dy) + 3) * PITCH + (dx) + 3] << 16) | (uint32_t)qlo[((dy) + 3) * PITCH + (dx) + 3]) ^ pm)
            // ring in the order of cv::FAST's 16-pattern: (0,3)(1,3)(2,2)(3,1)(3,0)(3,-1)(2,-2)(1,-3)(0,-3)(-1,-3)(-2,-2)(-3,-1)(-3,0)(-3,1)(-2,2)(-1,3)
            uint32_t r[16];
            r[0] = PX(0, 3);    r[1] = PX(1, 3);    r[2] = PX(2, 2);    r[3] = PX(3, 1);
            r[4] = PX(3, 0);    r[5] = PX(3, -1);   r[6] = PX(2, -2);   r[7] = PX(1, -3);
            r[8] = PX(0, -3);   r[9] = PX(-1, -3);  r[10] = PX(-2, -2); r[11] = PX(-3, -1);
            r[12] = PX(-3, 0);  r[13] = PX(-3, 1);  r[14] = PX(-2, 2);  r[15] = PX(-1, 3);
            const uint32_t ctr = PX(0, 0);
#undef PX
            uint32_t lo3[16], l9[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) lo3[k] = pk_min3_f16(r[k], r[(k + 1) & 15], r[(k + 2) & 15]);
#pragma unroll
            for (int k = 0; k < 16; ++k) l9[k] = pk_min3_f16(lo3[k], lo3[(k + 3) & 15], lo3[(k + 6) & 15]);
            const uint32_t m = pk_max_f16(
                pk_max3_f16(pk_max3_f16(l9[0], l9[1], l9[2]), pk_max3_f16(l9[3], l9[4], l9[5]), pk_max3_f16(l9[6], l9[7], l9[8])),
                pk_max3_f16(pk_max3_f16(l9[9], l9[10], l9[11]), pk_max3_f16(l9[12], l9[13], l9[14]), l9[15]));
            // (best arc minimum) - centre = score + 1 in both halves; <= t means "no corner at t"
            typedef short fast_s2 __attribute__((ext_vector_type(2)));
            const fast_s2 dd = __builtin_bit_cast(fast_s2, m) - __builtin_bit_cast(fast_s2, ctr);
            const int dlo = dd.x, dhi = dd.y;
            const bool tlo = dlo > th, thi = dhi > th;
            const bool clo = vlo && tlo, chi = vhi && thi;
            if (clo) sc[olo + (PITCH + SC_COL0)] = (uint8_t)dlo;   // score tile row y + 1, column x + SC_COL0; holds score + 1
            if (chi) sc[ohi + (PITCH + SC_COL0)] = (uint8_t)dhi;
            {   // the corners are also listed (in the keypoint array: NMS compacts it in place), so that NMS visits the
                // ~12 % of the pixels that are corners instead of all of them; a cell with more corners than the
                // array holds falls back to the pass over the whole score tile
                // (the masks of the compares themselves, cut to the lanes that hold an entry: a ballot of `clo` would be rebuilt from a select)
                const unsigned long long m0 = __builtin_amdgcn_ballot_w64(tlo) & vmlo, m1 = __builtin_amdgcn_ballot_w64(thi) & vmhi;
                const int n0 = __popcll(m0), add = n0 + __popcll(m1);
                if (ncorner + add <= kp_cap) {
                    const unsigned p0 = __builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u));
                    const unsigned p1 = __builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u));
                    if (clo) (s_kp + ncorner)[p0] = (unsigned short)elo;
                    if (chi) (s_kp + ncorner + n0)[p1] = (unsigned short)ehi;
                } else corner_overflow = true;
                ncorner += add;
            }
            nE = base;
            {   // both polarities passed the pre-test: back on the stack as "brighter" (the stack's top was just popped: the slots are free)
                const bool blo = vlo && (elo & 3u) == 3u, bhi = vhi && (ehi & 3u) == 3u;
                const unsigned long long q0 = __builtin_amdgcn_ballot_w64((elo & 3u) == 3u) & vmlo, q1 = __builtin_amdgcn_ballot_w64((ehi & 3u) == 3u) & vmhi;
                if ((q0 | q1) != 0ull) {
                    const int b0 = nE, b1 = nE + __popcll(q0);
                    nE = b1 + __popcll(q1);
                    if (blo) s_ent[b0 + __builtin_amdgcn_mbcnt_hi((unsigned)(q0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)q0, 0u))] = (unsigned short)(elo - 1u);
                    if (bhi) s_ent[b1 + __builtin_amdgcn_mbcnt_hi((unsigned)(q1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)q1, 0u))] = (unsigned short)(ehi - 1u);
                }
            }
            // the counters are wave-uniform; say so (a value merged behind one of the predicated stores above would otherwise count
            // as divergent and drag the loops that test it into exec-mask form)
            nE = __builtin_amdgcn_readfirstlane(nE);
            ncorner = __builtin_amdgcn_readfirstlane(ncorner);
            FD_STAMP(2);
        };

        unsigned e_base = (unsigned)(p_rsub * PITCH + p_x0) << 2;                  // entry of the lane's pixel 0, less its code
        const uint8_t* prow = tile + (p_rsub + 3) * PITCH + p_x0 + SC_COL0;         // the group's centre pixels
        for (int y0 = 0; y0 < sh; y0 += rps, e_base += (unsigned)(rps * PITCH) << 2, prow += rps * PITCH) {
            unsigned f6 = 0;     // byte j: which polarities pixel j of the lane's group passed in (1 = darker, 2 = brighter, 3 = both)
            if (p_active && y0 + p_rsub < sh) {
                const uint32_t* rowc = reinterpret_cast<const uint32_t*>(prow);
                const unsigned C = rowc[0], Dm = rowc[-1], Dp = rowc[1];
                const unsigned N = *reinterpret_cast<const uint32_t*>(prow - 3 * PITCH);
                const unsigned S = *reinterpret_cast<const uint32_t*>(prow + 3 * PITCH);
                const unsigned E = __builtin_amdgcn_alignbyte(Dp, C, 3);   // columns x0+3 .. x0+6
                const unsigned Wv = __builtin_amdgcn_alignbyte(C, Dm, 1);  // columns x0-3 .. x0
                const unsigned nC = ~C;
                const unsigned DN = __builtin_amdgcn_lerp(N, nC, rnd), DE = __builtin_amdgcn_lerp(E, nC, rnd);
                const unsigned DS = __builtin_amdgcn_lerp(S, nC, rnd), DW = __builtin_amdgcn_lerp(Wv, nC, rnd);
                const unsigned bN = __builtin_amdgcn_lerp(DN, kcB, 0u), bE = __builtin_amdgcn_lerp(DE, kcB, 0u);
                const unsigned bS = __builtin_amdgcn_lerp(DS, kcB, 0u), bW = __builtin_amdgcn_lerp(DW, kcB, 0u);
                const unsigned gN = __builtin_amdgcn_lerp(DN, kcD, 0u), gE = __builtin_amdgcn_lerp(DE, kcD, 0u);
                const unsigned gS = __builtin_amdgcn_lerp(DS, kcD, 0u), gW = __builtin_amdgcn_lerp(DW, kcD, 0u);
                // two adjacent compass points of one polarity: NE | ES | SW | WN == (N | S) & (E | W); v_bitop3 is full rate.
                // Darker, from the inverted flags g: (~gN | ~gS) & (~gE | ~gW) = ~(gN & gS) & ~(gE & gW)
                const unsigned pb = __builtin_amdgcn_bitop3_b32(bN | bS, bE, bW, 0xE0);
                const unsigned pd = __builtin_amdgcn_bitop3_b32(gN & gS, gE, gW, 0x07);
                // only the MSB of each byte of pb / pd is a flag (the lerps leave anything below it)
                f6 = (unsigned)__builtin_amdgcn_bitop3_b32(pd >> 1, p_vmask6, pb & p_vmask, 0xEA) >> 6;   // (a & b) | c
            }
            {   // append in RASTER order (lane = row, column group; a lane's pixels left to right): the lanes of a scoring pass then
                // gather their rings from neighbouring tile dwords -- appended slot by slot (one ballot per pixel slot, all lanes'
                // pixel 0 first) the same entries cost 2.4 x the LDS bank conflicts and 9 % of the kernel.  A lane counts its
                // pixels, a wave prefix sum gives its first slot, and each pixel that passed stores under its own exec mask (the
                // scalar instructions of the mask regions are free here: the kernel is bound by VECTOR issue)
                int total;
                const unsigned tb = (f6 | (f6 >> 1)) & 0x01010101u;        // byte j: 1 when pixel j passed
                const unsigned excl = (unsigned)wave_exclusive_scan(__popc(tb), total);
                if (total != 0) {
                    // byte j of g: 4 j + code = what pixel j adds to e_base; byte j of pf2: twice the number of the lane's pixels that
                    // passed before pixel j = the byte offset of pixel j's slot behind the lane's first (a 24-bit multiply-add: the
                    // multiplier's two terms carry pixel 0 into bytes 1, 2 and pixels 1, 2 into bytes 2, 3 / 3; tb << 25 adds pixel 0's
                    // share of byte 3).  Four masks under the full exec, then each pixel's store under its own; every byte pick
                    // is an SDWA operand select -- written out, because the compiler shifts and masks bytes 1 and 2 by hand
                    // (16 vector instructions for the append instead of 27).
#ifdef FAST_APPEND_C
                    {
                    const unsigned g = f6 + 0x0C080400u;
                    const unsigned pf2 = __umul24(tb, 0x020200u) + (tb << 25);
                    unsigned char* const slot0 = reinterpret_cast<unsigned char*>(s_ent + nE) + (excl << 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (((f6 >> (8 * j)) & 0xFFu) != 0u)
                            *reinterpret_cast<unsigned short*>(slot0 + ((pf2 >> (8 * j)) & 0xFFu)) = (unsigned short)(e_base + ((g >> (8 * j)) & 0xFFu));
                    }
#else
                    const unsigned g = f6 + 0x0C080400u, zero = 0u;
                    const unsigned slot0 = (unsigned)(size_t)(s_ent + nE) + (excl << 1);   // LDS byte address (s_ent + nE is scalar)
                    unsigned pf2, dat, adr;
                    unsigned long long sv, m0, m1, m2, m3;
                    asm volatile(
                        "v_lshlrev_b32 %[pf2], 25, %[tb]\n\t"
                        "v_mad_u32_u24 %[pf2], %[tb], %[k], %[pf2]\n\t"
                        "v_cmp_ne_u32_sdwa %[m0], %[f6], %[z] src0_sel:BYTE_0 src1_sel:DWORD\n\t"
                        "v_cmp_ne_u32_sdwa %[m1], %[f6], %[z] src0_sel:BYTE_1 src1_sel:DWORD\n\t"
                        "v_cmp_ne_u32_sdwa %[m2], %[f6], %[z] src0_sel:BYTE_2 src1_sel:DWORD\n\t"
                        "v_cmp_ne_u32_sdwa %[m3], %[f6], %[z] src0_sel:BYTE_3 src1_sel:DWORD\n\t"
                        "s_mov_b64 %[sv], exec\n\t"
                        "s_mov_b64 exec, %[m0]\n\t"
                        "v_add_u32_sdwa %[dat], %[eb], %[g] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
                        "ds_write_b16 %[s0], %[dat]\n\t"
                        "s_mov_b64 exec, %[m1]\n\t"
                        "v_add_u32_sdwa %[dat], %[eb], %[g] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
                        "v_add_u32_sdwa %[adr], %[s0], %[pf2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t"
                        "ds_write_b16 %[adr], %[dat]\n\t"
                        "s_mov_b64 exec, %[m2]\n\t"
                        "v_add_u32_sdwa %[dat], %[eb], %[g] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
                        "v_add_u32_sdwa %[adr], %[s0], %[pf2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\t"
                        "ds_write_b16 %[adr], %[dat]\n\t"
                        "s_mov_b64 exec, %[m3]\n\t"
                        "v_add_u32_sdwa %[dat], %[eb], %[g] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
                        "v_add_u32_sdwa %[adr], %[s0], %[pf2] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t"
                        "ds_write_b16 %[adr], %[dat]\n\t"
                        "s_mov_b64 exec, %[sv]"
                        : [pf2] "=&v"(pf2), [dat] "=&v"(dat), [adr] "=&v"(adr), [sv] "=&s"(sv), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
                        : [tb] "v"(tb), [k] "v"(0x020200u), [f6] "v"(f6), [z] "v"(zero), [eb] "v"(e_base), [g] "v"(g), [s0] "v"(slot0)
                        : "memory");
#endif
                    nE += total;
                }
            }
            const bool last = y0 + rps >= sh;
            while (nE >= 128 || (last && nE > 0)) score_pass(min(nE, 128));
        }
        wave_sync_lds();
        nkp = 0;
        if (!corner_overflow) {
            // NMS over the listed corners: strict > against the 8 neighbours' stored scores (0 below t / outside the scan
            // area); kept entries are compacted in place (a pass reads its 64 entries before any of them is overwritten)
            for (int i0 = 0; i0 < ncorner; i0 += WAVE) {
                const int i = i0 + lane;
                bool kept = false;
                unsigned e = 0;
                if (i < ncorner) {
                    e = s_kp[i];
                    const uint8_t* q = &sc[(e >> 2) + (PITCH + SC_COL0)];
                    const int v = q[0];
                    int nb = imax3(q[-PITCH - 1], q[-PITCH], q[-PITCH + 1]);
                    nb = imax3(nb, q[-1], q[1]);
                    nb = max(nb, imax3(q[PITCH - 1], q[PITCH], q[PITCH + 1]));
                    kept = v > nb;
                }
                wave_sync_lds();
                const unsigned long long mk = __ballot(kept);
                if (kept) s_kp[__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, (unsigned)nkp))] = (unsigned short)e;
                nkp += __popcll(mk);
            }
        } else
        // NMS byte-parallel over the whole score tile, 4 pixels per lane: n >= s per byte  <=>  MSB of (n + ~s + 1) >> 1.
        // A pixel below t holds 0 and is beaten by any neighbour; columns / rows around the scan area hold 0.
        for (int i0 = 0; i0 < nitems; i0 += WAVE) {
            const int it = i0 + lane;
            unsigned keep = 0, e0 = 0;
            if (it < nitems) {
                const int y = (int)(__umul24(it, inv_g) >> 20);
                const int x0 = (it - y * ngrp) * 4;
                e0 = (unsigned)(y * PITCH + x0) << 2;
                const uint32_t* s1 = reinterpret_cast<const uint32_t*>(&sc[(y + 1) * PITCH + x0 + SC_COL0]);
                const uint32_t* s0 = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(s1) - PITCH);
                const uint32_t* s2 = reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(s1) + PITCH);
                const uint32_t cur = s1[0];
                const uint32_t ns = ~cur, one = 0x01010101u;
                const uint32_t uc = s0[0], dc = s2[0];
                const uint32_t ul = __builtin_amdgcn_alignbyte(uc, s0[-1], 3), ur = __builtin_amdgcn_alignbyte(s0[1], uc, 1);
                const uint32_t cl = __builtin_amdgcn_alignbyte(cur, s1[-1], 3), cr = __builtin_amdgcn_alignbyte(s1[1], cur, 1);
                const uint32_t dl = __builtin_amdgcn_alignbyte(dc, s2[-1], 3), dr = __builtin_amdgcn_alignbyte(s2[1], dc, 1);
                const uint32_t g1 = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_lerp(ul, ns, one), __builtin_amdgcn_lerp(uc, ns, one), __builtin_amdgcn_lerp(ur, ns, one), 0xFE);
                const uint32_t g2 = __builtin_amdgcn_bitop3_b32(__builtin_amdgcn_lerp(dl, ns, one), __builtin_amdgcn_lerp(dc, ns, one), __builtin_amdgcn_lerp(dr, ns, one), 0xFE);
                const uint32_t g3 = __builtin_amdgcn_lerp(cl, ns, one) | __builtin_amdgcn_lerp(cr, ns, one);
                keep = ~(g1 | g2 | g3) & 0x80808080u;     // pixels beyond the scan row hold 0 and are never kept
            }
            // two strict maxima are never adjacent, so a group of 4 keeps at most 2 pixels: two ballots compact them.
            // The kept pixel's MSB is bit 8 j + 7: its entry is e0 + 4 j
            const int cnt = __popc(keep);
            const unsigned long long b1 = __ballot(cnt >= 1);
            if (b1) {
                const unsigned long long b2 = __ballot(cnt >= 2);
                const unsigned pos = __builtin_amdgcn_mbcnt_hi((unsigned)(b1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b1, (unsigned)nkp)) +
                                     __builtin_amdgcn_mbcnt_hi((unsigned)(b2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b2, 0u));
                if (cnt >= 1) s_kp[pos] = (unsigned short)(e0 + (((unsigned)(__ffs(keep) - 1) >> 1) & 0xCu));
                if (cnt >= 2) s_kp[pos + 1] = (unsigned short)(e0 + (((unsigned)(31 - __clz(keep)) >> 1) & 0xCu));
                nkp += __popcll(b1) + __popcll(b2);
            }
        }
        wave_sync_lds();
        FD_STAMP(3);
        if (nkp > 0 || minTh >= th) break;  // found corners, or the retry cannot find more
        th = minTh;
    }
    if (nkp == 0) { FD_FLUSH(sw * sh); return; }
    int base = 0;
    if (lane == 0) base = atomicAdd(&cand_count[(frame * nlevels + level) * ORB_CC_PAD], nkp);
    base = __shfl(base, 0, WAVE);
    unsigned long long* out = cand + L.cand_off + (size_t)frame * cand_frame_stride;
    for (int i = lane; i < nkp; i += WAVE) {
        const unsigned off = (unsigned)s_kp[i] >> 2;                  // y * PITCH + x
        const unsigned y = __umul24(off, (1u << 20) / PITCH + 1u) >> 20, x = off - y * PITCH;   // off < 2^13: the 20-bit reciprocal is exact
        const unsigned S = sc[off + (PITCH + SC_COL0)] - 1u;          // the tile holds score + 1
        const unsigned order = ((unsigned)c << 12) | ((unsigned)(y + 3) << 6) | (unsigned)(x + 3);
        const int o = base + i;
        if (o < L.cand_cap) out[o] = ((unsigned long long)S << 32) | order;
    }
    FD_STAMP(4);
    FD_FLUSH(sw * sh);
}

template <int PITCH>
__global__ __launch_bounds__(256) void fast_cells_kernel(
    const FastTab tab, const uint4* __restrict__ cells, int nlevels, int ncells,
    const uint8_t* __restrict__ img0, unsigned img0_stride, size_t img0_frame,
    const uint8_t* __restrict__ pyr,
    unsigned long long* __restrict__ cand, size_t cand_frame_stride,
    int* __restrict__ cand_count, int iniTh, int minTh, int tile_rows, int sc_rows, int kp_cap, int xcd_frames) {
    extern __shared__ __attribute__((aligned(16))) uint8_t fsm[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // A workgroup holds 1, 2 or 4 cells (waves).  One-cell workgroups: consecutive workgroup ids go to consecutive XCDs (8, an L2 each), but
    // neighbouring cells share image lines (6-pixel halos inside 128-byte lines) -- dealt cell by cell the launch fetched 545 MB instead of
    // 247 per 256 VGA frames.  So inside every run of 32 cells XCD x takes cells 4 x .. 4 x + 3 (what a four-wave workgroup held), and the
    // XCDs still walk the frame side by side.
    int cell, frame = blockIdx.y;
    if (xcd_frames) {   // 1-D grid, frames dealt to the XCDs: an XCD walks its frames one after the other, cell by cell
        const unsigned w = blockIdx.x, k = w >> 3, wpb = blockDim.x >> 6, P = ((unsigned)ncells + wpb - 1) / wpb, fq = k / P, local = k - fq * P;
        frame = (int)(8u * fq + ((w + fq) & 7u));   // (rotated per group of eight frames: an XCD does not keep meeting every eighth frame of a periodic input)
        if (frame >= xcd_frames) return;
        cell = (int)(local * wpb) + wv;
    } else if (blockDim.x == 64) {
        const unsigned w = blockIdx.x;
        cell = (int)((w & ~31u) + ((w & 7u) << 2) + ((w >> 3) & 3u));
    } else cell = blockIdx.x * (blockDim.x >> 6) + wv;
    if (cell >= ncells) return;
    fast_cell_wave<PITCH>(fsm, lane, wv, frame, cell, tab, cells, nlevels, img0, img0_stride, img0_frame, pyr, cand,
                          cand_frame_stride, cand_count, iniTh, minTh, tile_rows, sc_rows, kp_cap);
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

// exclusive scan of a[0..len) in place (LDS), returns the total; all NT threads call it
template <int NT>
__device__ int block_exclusive_scan(int* a, int len, int* wave_tmp /*[NT / 64]*/) {
    const int tid = threadIdx.x;
    const int C = (len + NT - 1) / NT;
    const int lo = min(tid * C, len), hi = min(lo + C, len);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += a[i];
    // inclusive scan of `sum` across the wave: four DPP row shifts scan each 16-lane row, two row broadcasts carry the row
    // totals on (lane 15 -> rows 1 and 3, lane 31 -> rows 2 and 3); shifted-out and masked-off lanes contribute the 0 of `old`.
    // (As six __shfl_up steps this was six dependent ds_bpermute round trips on the kernel's critical path.)
    int incl = sum;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, false);   // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, false);   // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, false);   // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, false);   // row_shr:8
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);   // row_bcast:15 into rows 1, 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);   // row_bcast:31 into rows 2, 3
    if ((tid & 63) == 63) wave_tmp[tid >> 6] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const int v = wave_tmp[w];
        wbase += w < (tid >> 6) ? v : 0;
        total += v;
    }
    int run = wbase + incl - sum;
    for (int i = lo; i < hi; ++i) {
        int v = a[i];
        a[i] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

#ifdef OCT_DIAG   // diagnostic builds only: per (frame, level) workgroup phase ticks
__device__ unsigned long long g_oct_ph[4096 * 8];
extern "C" int slamit_diag_oct(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_oct_ph), sizeof(unsigned long long) * 4096 * 8); }
#define OD_DECL unsigned long long od_ph[6] = {0, 0, 0, 0, 0, 0}, od_prev = __builtin_amdgcn_s_memtime(), od_rt0 = __builtin_amdgcn_s_memrealtime(); int od_pass = 0
#define OD_STAMP(i) do { unsigned long long tn = __builtin_amdgcn_s_memtime(); od_ph[i] += tn - od_prev; od_prev = tn; } while (0)
#else
#define OD_DECL
#define OD_STAMP(i)
#endif

#ifndef OCT_THREADS
#define OCT_THREADS 512
#endif
__host__ __device__ inline size_t orbk_octree_node_bytes(int node_cap) {
    return ((size_t)node_cap * (8 + 2 * 8 + 2 * 4 + 4 * 4 + 5 * 4 + 4 + 4) + 64 + 15) & ~(size_t)15;
}
#define OCT_LDS_KEYS_MAX 10240   // LDS key arrays hold at most this many candidates of one (frame, level), 6 bytes each
#define OCT_LDS_KEYS_MIN 2048
#define OCT_LDS_BUDGET (78 * 1024)   // per workgroup, so that two fit a CU

// depth of the path table the full passes are served from: the deepest of 4, 3, 2 whose table (4 + .. + 4^depth ints per root)
// fits `room` ints; 0 = no table (a sweep over the keys per pass)
__device__ __forceinline__ int nIniFD(int nIni, int room) {
    return nIni * 340 <= room ? 4 : nIni * 84 <= room ? 3 : nIni * 20 <= room ? 2 : 0;
}

// XY / ND: per-candidate position and current node, in LDS when the list fits (LK) and in the HBM workspace otherwise.
template <int NT, bool LK>
__device__ __forceinline__ void octree_body(
    const OrbLevel& L, const unsigned long long* __restrict__ K, int n_keys, uint32_t* XY, uint16_t* ND,
    OrbLevelKp* __restrict__ OUT, int* __restrict__ kp_count_out, unsigned char* smem, int cap, int* wave_tmp, int* s_vars) {
    const int tid = threadIdx.x;
    // LDS carve (all 8-byte aligned)
    unsigned long long* best = reinterpret_cast<unsigned long long*>(smem);              // cap
    // the two generations of node boxes / populations are addressed as base + generation * cap: an array of two pointers
    // indexed at run time loses the LDS address space and every access becomes a FLAT load (slower, and it counts in
    // both wait counters)
    Box16* const box0 = reinterpret_cast<Box16*>(best + cap);   // 2 * cap
    int* const cnt0 = reinterpret_cast<int*>(box0 + 2 * cap);    // 2 * cap
    int* childcnt = cnt0 + 2 * cap;      // 4*cap
    int* scanbuf = childcnt + 4 * cap;   // 5*cap
    int* rankv = scanbuf + 5 * cap;      // cap   (rank of an expandable node in the sorted order)
    int* sorted = rankv + cap;           // cap   (node index at each rank)
    int& s_n = s_vars[0]; int& s_expand = s_vars[1]; int& s_k = s_vars[2]; int& s_flag = s_vars[3];
    const int N = L.quota;
    OD_DECL;
    // The passes that split EVERY expandable node ("full" passes, ORBextractor.cc:600-686) do not need a sweep over the keys each:
    // a key's quadrant path down to depth FD follows from its root box alone, so ONE sweep histograms the depth-FD paths, the
    // shallower populations are sums of four, and a full pass takes its children's populations from the table (node i carries
    // root | depth | path).  The keys get their node labels once, when the full passes end, through a path -> node table.
    // The histogram lives in `best` (used after the loop only), the node codes in rankv/sorted (used by "largest first" passes,
    // i.e. after the full ones).  Depth 4 when the table fits (340 ints per root), else 3, 2 or the sweep per pass.
    const int two_cap = 2 * cap;
    const int FD = nIniFD(L.nIni, two_cap);
    const int HS = ((4 << (2 * FD)) - 4) / 3;          // table ints per root: 4 + 16 + ... + 4^FD
    int* const hist = reinterpret_cast<int*>(best);
    int* const ncode0 = rankv;                          // [2 * cap]: two generations, like box0 / cnt0

    // ---- roots (ORBextractor.cc:556-598) ----
    const int nIni = L.nIni;
    if (tid < nIni) {
        Box16 b; b.x0 = (short)L.rootUL[tid]; b.y0 = 0; b.x1 = (short)L.rootUR[tid]; b.y1 = (short)L.boxH;
        box0[tid] = b;
        cnt0[tid] = 0;
    }
    for (int e = tid; e < nIni * HS; e += NT) hist[e] = 0;
    __syncthreads();
    {
        const int nCols = L.nCols, wCell = L.wCell, hCell = L.hCell;
        const unsigned inv_cols = 0xFFFFFFFFu / (unsigned)nCols + 1u;   // cell / nCols == umulhi(cell, inv) while cell * nCols < 2^32
        const float hX = L.hX;
        const int lane = tid & 63;
        // (every lane stays in the loop: the run lengths below are taken over whole waves; four keys per thread are fetched before
        // the first is used -- the loop body's LDS atomics keep the compiler from overlapping the trips' global loads itself)
        for (int kb4 = 0; kb4 < n_keys; kb4 += 4 * NT) {
            unsigned ord4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) ord4[j] = (unsigned)K[min(kb4 + j * NT + tid, n_keys - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
            const int kb = kb4 + j * NT;
            if (kb >= n_keys) break;
            const int k = kb + tid;
            const bool valid = k < n_keys;
            unsigned order = ord4[j];
            int lx = order & 63, ly = (order >> 6) & 63, cell = order >> 12;
            int ci = (int)__umulhi((unsigned)cell, inv_cols), cj = cell - ci * nCols;
            int x = cj * wCell + lx, y = ci * hCell + ly;
            int r = (int)((float)x / hX);  // vpIniNodes[kp.pt.x/hX]
            r = min(r, nIni - 1);
            if (valid) XY[k] = (uint32_t)x | ((uint32_t)y << 16);
            if (FD) {
                Box16 b = box0[r];
                int path = 0;
                for (int d = 0; d < FD; ++d) {
                    int mx, my;
                    const int q = box_quadrant(b, x, y, &mx, &my);
                    b = child_box(b, q);
                    path = path * 4 + q;
                }
                if (valid) ND[k] = (uint16_t)((r << 8) | path);
                // the keys come in cell order, so neighbouring lanes mostly share a bin: one atomic per RUN of equal bins
                const int bin = valid ? r * HS + (HS - (1 << (2 * FD))) + path : -1;
                const int prev = __shfl_up(bin, 1);
                const bool head = lane == 0 || bin != prev;
                const unsigned long long heads = __ballot(head);
                if (head && bin >= 0) {
                    const unsigned long long rest = (heads >> lane) >> 1;
                    atomicAdd(&hist[bin], rest ? (int)__builtin_ctzll(rest) + 1 : 64 - lane);
                }
            } else {
                if (valid) ND[k] = (uint16_t)r;
                // every key of a wave falls into one of <= 8 roots: count per root with ballots, one atomic per wave and
                // root (same-address LDS atomics of 64 lanes would serialise)
                for (int q = 0; q < nIni; ++q) {
                    const unsigned long long m = __ballot(valid && r == q);
                    if (m && lane == (int)__builtin_ctzll(m)) atomicAdd(&cnt0[q], (int)__popcll(m));
                }
            }
            }
        }
    }
    __syncthreads();
    for (int d = FD - 1; d >= 1; --d) {   // populations of the shallower paths
        const int off = ((4 << (2 * d)) - 4) / 3 - (1 << (2 * d)), offc = off + (1 << (2 * d)), per = 1 << (2 * d);
        for (int e = tid; e < nIni * per; e += NT) {
            const int r = e >> (2 * d), pth = e & (per - 1);
            const int* c = &hist[r * HS + offc + 4 * pth];
            hist[r * HS + off + pth] = (c[0] + c[1]) + (c[2] + c[3]);
        }
        __syncthreads();
    }
    if (tid == 0) {  // erase empty roots, keep order (<= 8 roots)
        int m = 0, dropped = 0;
        for (int i = 0; i < nIni; ++i) {
            if (FD) cnt0[i] = (hist[i * HS] + hist[i * HS + 1]) + (hist[i * HS + 2] + hist[i * HS + 3]);
            if (cnt0[i] > 0) { scanbuf[i] = m; box0[m] = box0[i]; cnt0[m] = cnt0[i]; ncode0[m] = i << 16; ++m; }
            else { scanbuf[i] = -1; dropped = 1; }
        }
        s_n = m; s_flag = dropped;
    }
    __syncthreads();
    if (s_flag && !FD) {
        for (int k = tid; k < n_keys; k += NT) ND[k] = (uint16_t)scanbuf[ND[k]];
        __syncthreads();
    }

    int cur = 0;
    int n = s_n;
    bool finish = false, careful = false;
    bool tabled = FD > 0;     // the keys still carry root | path; node populations come from the table
    int depth_done = 0;
    // keys -> nodes once the full passes are over: every node of the list enters its index at its (root, depth, path); a key
    // looks its path up from the deepest level to the root (the list's nodes are disjoint, so exactly one level answers)
    auto label_keys = [&]() {
        for (int e = tid; e < nIni * HS; e += NT) hist[e] = -1;
        if (tid < ORB_MAX_ROOTS) scanbuf[tid] = -1;
        __syncthreads();
        for (int i = tid; i < n; i += NT) {
            const int c = ncode0[cur * cap + i], r = c >> 16, d = (c >> 12) & 15, pth = c & 4095;
            if (d == 0) scanbuf[r] = i;
            else hist[r * HS + ((4 << (2 * d)) - 4) / 3 - (1 << (2 * d)) + pth] = i;
        }
        __syncthreads();
        for (int k = tid; k < n_keys; k += NT) {
            const int lab = ND[k], r = lab >> 8, pth = lab & 255;
            int node = scanbuf[r];
            for (int d = 1; d <= FD; ++d) {
                const int v = hist[r * HS + ((4 << (2 * d)) - 4) / 3 - (1 << (2 * d)) + (pth >> (2 * (FD - d)))];
                node = v >= 0 ? v : node;
            }
            ND[k] = (uint16_t)node;
        }
        __syncthreads();
    };
    OD_STAMP(0);
    while (!finish) {
        const int prevSize = n;
#ifdef OCT_DIAG
        ++od_pass;
#endif
        const bool tpass = tabled && !careful && depth_done < FD;   // a full pass served from the table
        if (tabled && !tpass) { label_keys(); tabled = false; }
        if (tpass) {
            for (int e = tid; e < 4 * n; e += NT) {
                const int i = e >> 2, c = ncode0[cur * cap + i], r = c >> 16, d = (c >> 12) & 15, pth = c & 4095;
                childcnt[e] = cnt0[cur * cap + i] > 1 ? hist[r * HS + ((4 << (2 * (d + 1))) - 4) / 3 - (1 << (2 * (d + 1))) + 4 * pth + (e & 3)] : 0;
            }
        } else {
        // children populations of every expandable node
        for (int i = tid; i < 4 * n; i += NT) childcnt[i] = 0;
        __syncthreads();
        for (int k = tid; k < n_keys; k += NT) {
            int nd = ND[k];
            if (cnt0[cur * cap + nd] > 1) {
                uint32_t xy = XY[k];
                int mx, my;
                int q = box_quadrant(box0[cur * cap + nd], xy & 0xFFFF, xy >> 16, &mx, &my);
                atomicAdd(&childcnt[nd * 4 + q], 1);
                ND[k] = (uint16_t)(nd | ((q + 1) << 12));   // the quadrant rides in the label's top bits until the re-label sweep
            }
        }
        }
        __syncthreads();
        OD_STAMP(1);

        int kproc;  // number of parents split in this pass, in processing order
        if (!careful) {
            // every expandable node is split, in list order
            kproc = -1;
        } else {
            // "largest first": rank expandable nodes by (population desc, list position asc)
            // one wavefront per node i, lanes over j: rank = number of expandable nodes that come before i
            for (int i = tid >> 6; i < n; i += NT / 64) {
                const int ci_ = cnt0[cur * cap + i];
                int r = -1;
                if (ci_ > 1) {
                    r = 0;
                    for (int j0 = 0; j0 < n; j0 += 64) {
                        const int j = j0 + (tid & 63);
                        const int cj_ = j < n ? cnt0[cur * cap + j] : 0;
                        r += (int)__popcll(__ballot((cj_ > 1) && (cj_ > ci_ || (cj_ == ci_ && j < i))));
                    }
                    if ((tid & 63) == 0) sorted[r] = i;
                }
                if ((tid & 63) == 0) rankv[i] = r;
            }
            if (tid == 0) s_expand = 0;
            __syncthreads();
            // m = number of expandable nodes
            int local = 0;
            for (int i = tid; i < n; i += NT) local += cnt0[cur * cap + i] > 1;
            if (local) atomicAdd(&s_expand, local);
            __syncthreads();
            const int m = s_expand;
            // gains in sorted order, prefix, first position where the list reaches N
            for (int s = tid; s < m; s += NT) {
                int i = sorted[s];
                const int* cc = &childcnt[i * 4];
                scanbuf[s] = (cc[0] > 0) + (cc[1] > 0) + (cc[2] > 0) + (cc[3] > 0) - 1;
            }
            if (tid == 0) s_k = m;
            __syncthreads();
            block_exclusive_scan<NT>(scanbuf, m, wave_tmp);  // scanbuf[s] = gain of ranks < s
            for (int s = tid; s < m; s += NT) {
                int i = sorted[s];
                const int* cc = &childcnt[i * 4];
                int g = (cc[0] > 0) + (cc[1] > 0) + (cc[2] > 0) + (cc[3] > 0) - 1;
                if (n + scanbuf[s] + g >= N) atomicMin(&s_k, s + 1);
            }
            __syncthreads();
            kproc = s_k;
        }

        OD_STAMP(2);
        // flags of the new list: children of split parents in reverse processing order (n4..n1),
        // then the nodes that stay, in their old order
        int nchildslots;
        if (!careful) {
            nchildslots = 4 * n;
            for (int e = tid; e < 4 * n; e += NT) {
                int i = n - 1 - (e >> 2), q = 3 - (e & 3);
                scanbuf[e] = (cnt0[cur * cap + i] > 1) && (childcnt[i * 4 + q] > 0);
            }
            for (int i = tid; i < n; i += NT) scanbuf[4 * n + i] = cnt0[cur * cap + i] <= 1;
        } else {
            nchildslots = 4 * kproc;
            for (int e = tid; e < nchildslots; e += NT) {
                int s = kproc - 1 - (e >> 2), q = 3 - (e & 3);
                scanbuf[e] = childcnt[sorted[s] * 4 + q] > 0;
            }
            for (int i = tid; i < n; i += NT) {
                int r = rankv[i];
                scanbuf[nchildslots + i] = !(r >= 0 && r < kproc);
            }
        }
        __syncthreads();
        const int n_new = block_exclusive_scan<NT>(scanbuf, nchildslots + n, wave_tmp);
        // (n_new <= cap by construction: a full pass only runs while n + 3*expandable <= N, the
        //  largest-first pass stops within 3 of N; clamp anyway for memory safety)
        const int nxt = cur ^ 1;
        if (tid == 0) s_expand = 0;
        __syncthreads();
        int local_expand = 0;
        for (int e = tid; e < nchildslots; e += NT) {
            int i, q;
            if (!careful) { i = n - 1 - (e >> 2); q = 3 - (e & 3); }
            else { i = sorted[kproc - 1 - (e >> 2)]; q = 3 - (e & 3); }
            bool split = !careful ? (cnt0[cur * cap + i] > 1) : true;
            int cc = childcnt[i * 4 + q];
            if (split && cc > 0) {
                int pos = scanbuf[e];
                if (pos < cap) {
                    box0[nxt * cap + pos] = child_box(box0[cur * cap + i], q);
                    cnt0[nxt * cap + pos] = cc;
                    if (tpass) { const int c = ncode0[cur * cap + i]; ncode0[nxt * cap + pos] = (c & ~0xFFFF) | ((((c >> 12) & 15) + 1) << 12) | (4 * (c & 4095) + q); }
                }
                local_expand += cc > 1;
            }
        }
        for (int i = tid; i < n; i += NT) {
            bool stays = !careful ? (cnt0[cur * cap + i] <= 1) : !(rankv[i] >= 0 && rankv[i] < kproc);
            if (stays) {
                int pos = scanbuf[nchildslots + i];
                if (pos < cap) {
                    box0[nxt * cap + pos] = box0[cur * cap + i]; cnt0[nxt * cap + pos] = cnt0[cur * cap + i];
                    if (tpass) ncode0[nxt * cap + pos] = ncode0[cur * cap + i];
                }
            }
        }
        if (local_expand) atomicAdd(&s_expand, local_expand);
        OD_STAMP(3);
        // re-label the keys (not while the table serves the passes: the keys keep root | path)
        if (!tpass)
        for (int k = tid; k < n_keys; k += NT) {
            const int lab = ND[k];
            const int nd = lab & 4095, q = (lab >> 12) - 1;   // q >= 0 exactly for keys of expandable nodes (cnt > 1)
            bool split = !careful ? (q >= 0) : (rankv[nd] >= 0 && rankv[nd] < kproc);
            int pos;
            if (split) {
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
        ++depth_done;
        OD_STAMP(4);
    }
    if (tabled) label_keys();

    // ---- best key per node (ORBextractor.cc:755-773): max response, first in input order ----
    for (int i = tid; i < n; i += NT) best[i] = 0ull;
    __syncthreads();
    for (int k = tid; k < n_keys; k += NT) {
        unsigned long long c = K[k];
        unsigned long long packed = (c & 0xFFFFFFFF00000000ull) | (0xFFFFFFFFu - (unsigned)c);
        atomicMax(&best[ND[k]], packed);
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        unsigned long long b = best[i];
        unsigned order = 0xFFFFFFFFu - (unsigned)b;
        int lx = order & 63, ly = (order >> 6) & 63, cell = order >> 12;
        int ci = cell / L.nCols, cj = cell - ci * L.nCols;
        OrbLevelKp kp;
        kp.x = (int16_t)(cj * L.wCell + lx + ORB_MIN_BORDER);  // ORBextractor.cc:863-864
        kp.y = (int16_t)(ci * L.hCell + ly + ORB_MIN_BORDER);
        kp.response = (float)(unsigned)(b >> 32);
        kp.angle = 0.f; kp.cs = 1.f; kp.sn = 0.f;
        if (i < L.kp_cap) OUT[i] = kp;
    }
    if (tid == 0) *kp_count_out = min(n, L.kp_cap);
#ifdef OCT_DIAG
    OD_STAMP(5);
    if (tid == 0) {
        const int w = blockIdx.y * gridDim.x + blockIdx.x;
        if (w < 4096) { unsigned long long* o = g_oct_ph + 8 * w; for (int k = 0; k < 6; ++k) o[k] = od_ph[k]; o[6] = ((unsigned long long)od_pass << 32) | (unsigned)n_keys; o[7] = od_rt0; o[5] = __builtin_amdgcn_s_memrealtime(); }
    }
#endif
}

template <int OCT_NT>
__global__ __launch_bounds__(OCT_NT) void octree_kernel(
    const OrbLevel* __restrict__ levels, int nlevels,
    const unsigned long long* __restrict__ cand, size_t cand_frame_stride,
    int* __restrict__ cand_count,
    uint32_t* __restrict__ ws_xy, uint16_t* __restrict__ ws_node,
    OrbLevelKp* __restrict__ lkp, size_t kp_frame_stride, int* __restrict__ kp_count,
    int node_cap, int key_cap, int level_override /* -1: blockIdx.x */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // node arrays | k_xy[key_cap] | k_nd[key_cap]
    __shared__ int wave_tmp[OCT_NT / 64];
    __shared__ int s_vars[4];
    uint32_t* k_xy = reinterpret_cast<uint32_t*>(smem + orbk_octree_node_bytes(node_cap));
    uint16_t* k_nd = reinterpret_cast<uint16_t*>(k_xy + key_cap);
    // grid = (frames, levels): workgroups are dealt in x-major order, so every frame's level 0 (the longest list, the
    // longest-running workgroup) starts first and a batch that needs a second round of workgroups fills it with the short
    // top levels -- with (levels, frames) 128 frames took two full rounds (96 us instead of 52)
    const int level = level_override >= 0 ? level_override : blockIdx.y;
    const int frame = blockIdx.x;
    const OrbLevel& L = levels[level];
    const int kidx = frame * nlevels + level;
    const int n_raw = cand_count[kidx * ORB_CC_PAD];
    const int n_keys = min(n_raw, L.cand_cap);
    // this workgroup is the counter's only reader: hand it back zeroed for the next call's FAST pass (no memset launch
    // per call) and keep the count in the line's second word for slamit_orb_debug_candidates
    __syncthreads();
    if (threadIdx.x == 0) { cand_count[kidx * ORB_CC_PAD] = 0; cand_count[kidx * ORB_CC_PAD + 1] = n_raw; }
    const unsigned long long* K = cand + L.cand_off + (size_t)frame * cand_frame_stride;
    OrbLevelKp* OUT = lkp + L.kp_off + (size_t)frame * kp_frame_stride;
    if (n_keys == 0) {
        if (threadIdx.x == 0) kp_count[kidx] = 0;
        return;
    }
    if (n_keys <= key_cap)
        octree_body<OCT_NT, true>(L, K, n_keys, k_xy, k_nd, OUT, &kp_count[kidx], smem, node_cap, wave_tmp, s_vars);
    else
        octree_body<OCT_NT, false>(L, K, n_keys, ws_xy + L.cand_off + (size_t)frame * cand_frame_stride,
                                        ws_node + L.cand_off + (size_t)frame * cand_frame_stride, OUT, &kp_count[kidx], smem,
                                        node_cap, wave_tmp, s_vars);
}

// --------------------------------------------------------------------------------------------
// K5: IC_Angle — intensity centroid over the radius-15 disc on the un-blurred level.
// One wavefront per keypoint; integer moments reduced with cross-lane shuffles.
// --------------------------------------------------------------------------------------------
// offsets (u, v) of the 749 pixels of the radius-15 disc, rows v = -15..15, |u| <= umax[|v|]
// per LANE: its 12 disc pixels (index lane + 64 j) as (u, v) byte pairs, 24 bytes in a row: a lane fetches them with two loads
// (dwordx4 + dwordx2) instead of 24 byte loads -- a wave of this kernel only handles four keypoints, so its table loads are a
// third of its vector-memory instructions
struct DiscTable { signed char uv[64][32]; int n; };   // [lane][2 j], [lane][2 j + 1] = u, v; 8 bytes of padding per lane
__constant__ DiscTable c_disc;

static DiscTable make_disc() {
    static const int umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};
    DiscTable t;
    for (int l = 0; l < 64; ++l) for (int k = 0; k < 32; ++k) t.uv[l][k] = 0;   // padding entries are (0, 0): weight zero
    t.n = 0;
    for (int v = -15; v <= 15; ++v)
        for (int u = -umax[v < 0 ? -v : v]; u <= umax[v < 0 ? -v : v]; ++u) {
            t.uv[t.n & 63][2 * (t.n >> 6)] = (signed char)u; t.uv[t.n & 63][2 * (t.n >> 6) + 1] = (signed char)v;
            ++t.n;
        }
    return t;
}

#define IC_KP_PER_WAVE 8
#define IC_R 15                   // HALF_PATCH_SIZE
#define IC_ROWS (2 * IC_R + 1)
#define IC_PITCH 32               // 31 columns in two 16-byte chunks

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
__device__ __forceinline__ int wave_sum_i32(int v) {   // the same value in every lane; no LDS round trips (ds_bpermute) involved
    v += dpp_i32<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_i32<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_i32<0x141>(v);   // row_half_mirror
    v += dpp_i32<0x140>(v);   // row_mirror
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

__global__ __launch_bounds__(256) void ic_angle_kernel(
    const OrbLevel* __restrict__ levels, int nlevels,
    const uint8_t* __restrict__ img0, size_t img0_stride, size_t img0_frame,
    const uint8_t* __restrict__ pyr,
    OrbLevelKp* __restrict__ lkp, size_t kp_frame_stride, const int* __restrict__ kp_count, int xcd_frames, int kblocks) {
    const int lane = threadIdx.x & 63;
    int level, frame, kb;
    if (xcd_frames) {   // 1-D grid, frames dealt to the XCDs (see describe_kernel)
        const unsigned w = blockIdx.x, k = w >> 3, P = (unsigned)(nlevels * kblocks), fq = k / P, local = k - fq * P;
        frame = (int)(8u * fq + ((w + fq) & 7u));   // (rotated per group of eight frames: an XCD does not keep meeting every eighth frame of a periodic input)
        if (frame >= xcd_frames) return;
        level = (int)(local / (unsigned)kblocks); kb = (int)(local - (unsigned)level * (unsigned)kblocks);
    } else { level = blockIdx.y; frame = blockIdx.z; kb = blockIdx.x; }
    const OrbLevel& L = levels[level];
    const int count = kp_count[frame * nlevels + level];
    const int i0 = (kb * 4 + (threadIdx.x >> 6)) * IC_KP_PER_WAVE;
    if (i0 >= count) return;
    const uint8_t* src;
    int stride;
    if (level == 0) { src = img0 + (size_t)frame * img0_frame; stride = (int)img0_stride; }
    else { src = pyr + L.plane_off + (size_t)frame * L.plane_bytes; stride = L.stride; }
    // The wave's keypoints at once.  A keypoint's 31 x 31 window is STAGED in LDS with one 16-byte load per lane (31 rows x two
    // chunks = 62 lanes; the loads start at the window's own first column, whatever its alignment), then the lane's 12 disc pixels
    // are LDS byte reads at constant offsets.  Fetching the 749 disc bytes straight from the plane took 12 byte-gather
    // instructions per keypoint, each a 64-address trip through the texture addresser: that unit, not memory, bound the pass
    // (3.07 M gathers per 256 frames x 16 cycles over 256 CUs = 80 of its 90 us).  The atan2 / sincos tail (~150 instructions)
    // runs ONCE with keypoint k on lane k instead of once per keypoint on lane 0.
    __shared__ __attribute__((aligned(16))) uint8_t s_win[4][IC_KP_PER_WAVE][IC_ROWS * IC_PITCH];
    const int wvi = threadIdx.x >> 6;
    OrbLevelKp* kp0 = lkp + L.kp_off + (size_t)frame * kp_frame_stride;
    typedef unsigned ic_u4 __attribute__((ext_vector_type(4), aligned(1)));
    {
        const int lr = min(lane >> 1, IC_ROWS - 1), lc = lane & 1;
        const unsigned loff = __umul24((unsigned)lr, (unsigned)stride) + 16u * (unsigned)lc;
        ic_u4 ld[IC_KP_PER_WAVE];
#pragma unroll
        for (int k = 0; k < IC_KP_PER_WAVE; ++k) {
            const int i = min(i0 + k, count - 1);
            const uint8_t* corner = src + (__umul24((unsigned)(kp0[i].y - IC_R), (unsigned)stride) + (unsigned)(kp0[i].x - IC_R));
            ld[k] = *reinterpret_cast<const ic_u4*>(corner + loff);
        }
#pragma unroll
        for (int k = 0; k < IC_KP_PER_WAVE; ++k)
            if (lane < 2 * IC_ROWS) *reinterpret_cast<uint4*>(&s_win[wvi][k][lr * IC_PITCH + 16 * lc]) = make_uint4(ld[k].x, ld[k].y, ld[k].z, ld[k].w);
    }
    // this lane's 12 disc pixels (749 = 11 * 64 + 45): window offsets and weights live in registers
    int off[12], wu[12], wv[12];
    {
        const uint4 t0 = *reinterpret_cast<const uint4*>(&c_disc.uv[lane][0]);
        const uint2 t1 = *reinterpret_cast<const uint2*>(&c_disc.uv[lane][16]);
        const uint32_t tw[6] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y};
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const uint32_t pr = tw[j >> 1] >> (16 * (j & 1));
            const int u = (int)(signed char)(pr & 0xFFu), v = (int)(signed char)((pr >> 8) & 0xFFu);
            off[j] = (v + IC_R) * IC_PITCH + (u + IC_R); wu[j] = u; wv[j] = v;
        }
    }
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    wave_sync_lds();
    int val[IC_KP_PER_WAVE][12];
#pragma unroll
    for (int k = 0; k < IC_KP_PER_WAVE; ++k)
#pragma unroll
        for (int j = 0; j < 12; ++j) val[k][j] = s_win[wvi][k][off[j]];
    // The 2 x IC_KP_PER_WAVE = 16 moments are summed over the wave TOGETHER (integers: any order is exact): a butterfly over lane
    // bits 0 .. 3 that halves the number of values at every stage (a lane keeps the value its bit selects and takes the
    // partner lane's partial of it), so a lane ends with the 16-lane-row partial of moment number lane & 15; the four rows meet
    // in LDS.  71 vector operations instead of 16 x 12 for sixteen separate wave sums.
    int V[2 * IC_KP_PER_WAVE];
#pragma unroll
    for (int k = 0; k < IC_KP_PER_WAVE; ++k) {
        int m10 = 0, m01 = 0;
#pragma unroll
        for (int j = 0; j < 12; ++j) { m10 += __mul24(wu[j], val[k][j]); m01 += __mul24(wv[j], val[k][j]); }
        V[2 * k] = m10; V[2 * k + 1] = m01;
    }
    {
        const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4, b3 = lane & 8;
#pragma unroll
        for (int p = 0; p < 8; ++p) { const int a = V[2 * p], b = V[2 * p + 1]; V[p] = (b0 ? b : a) + dpp_i32<0xB1>(b0 ? a : b); }            // partner lane ^ 1
#pragma unroll
        for (int p = 0; p < 4; ++p) { const int a = V[2 * p], b = V[2 * p + 1]; V[p] = (b1 ? b : a) + dpp_i32<0x4E>(b1 ? a : b); }            // lane ^ 2
#pragma unroll
        for (int p = 0; p < 2; ++p) { const int a = V[2 * p], b = V[2 * p + 1]; V[p] = (b2 ? b : a) + dpp_i32<0x1B>(dpp_i32<0x141>(b2 ? a : b)); }   // lane ^ 4: half-row mirror, then quad reverse
        { const int a = V[0], b = V[1]; V[0] = (b3 ? b : a) + dpp_i32<0x141>(dpp_i32<0x140>(b3 ? a : b)); }                                     // lane ^ 8: row mirror, then half-row mirror
    }
    __shared__ int s_rows[4][64];
    int* sr = s_rows[threadIdx.x >> 6];
    sr[lane] = V[0];
    wave_sync_lds();
    const int tot = (sr[lane & 15] + sr[(lane & 15) + 16]) + (sr[(lane & 15) + 32] + sr[(lane & 15) + 48]);   // moment number lane & 15: m10 of keypoint (lane & 15) >> 1 on even lanes, m01 on odd
    const int my10 = tot, my01 = dpp_i32<0xB1>(tot);   // (meaningful on the even lanes 0 .. 14: keypoint lane >> 1)
    if (lane < 2 * IC_KP_PER_WAVE && !(lane & 1) && i0 + (lane >> 1) < count) {
        const float ang = slamit_fast_atan2((float)my01, (float)my10);
        float sn, cs;
        slamit_sincosf(ang * factorPI, &sn, &cs);   // a = cos, b = sin of computeOrbDescriptor, once per keypoint
        OrbLevelKp* kp = kp0 + i0 + (lane >> 1);
        kp->angle = ang; kp->cs = cs; kp->sn = sn;
    }
}

// --------------------------------------------------------------------------------------------
// K6: 7x7 Gaussian blur, fixed point taps [18,34,49,55,49,34,18] (sum 257), REFLECT_101 on the
// un-padded level.  Workgroup = 64x16 output tile; rows pass kept as u16 in LDS (<= 255*257).
// --------------------------------------------------------------------------------------------
// BORDER_REFLECT_101 for the coordinates the blur can produce: at most one reflection is ever needed for a pixel
// that contributes to an output (|overshoot| <= 3 and every level is >= 62 pixels wide and tall, slamit_orb_create
// refuses smaller ones); coordinates further out (only staged, never used by an in-image output) are clamped so
// that their address stays inside the plane.  Branch-free.
__device__ __forceinline__ int reflect101(int p, int len) {
    p = p < 0 ? -p : p;
    p = p >= len ? 2 * len - 2 - p : p;
    return min(max(p, 0), len - 1);
}

// One launch covers every level of every frame: blockIdx.x walks the per-level tile lists
// (OrbLevel::blur_tile_base), blockIdx.y is the frame.  A workgroup owns a 64-wide, 64-tall strip
// and walks it in four 16-row steps; the input rows of step s+1 are fetched into registers while
// step s is being filtered, so the HBM/L2 latency is paid once per strip.  Dwords that lie fully
// inside an image row are loaded whole (planes are 4-byte aligned); edge dwords go byte by byte
// through the REFLECT_101 map.
#define BLUR_STEPS 4

typedef unsigned short blur_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) {   // a.lo*b.lo + a.hi*b.hi + c (v_dot2_u32_u16)
    return __builtin_amdgcn_udot2(__builtin_bit_cast(blur_us2, a), __builtin_bit_cast(blur_us2, b), c, false);
}

// One staged dword of a step: columns gx0 .. gx0+3 of image row (by + r - 3), REFLECT_101 at the borders.  Everything
// that does not depend on the step (r, gx0, whether the dword lies inside the row) is computed once per thread.
__device__ __forceinline__ uint32_t blur_fetch(const uint8_t* S, unsigned sstride, bool whole, int w, int h, int gx0, int row) {
    const unsigned ro = (unsigned)reflect101(row, h) * sstride;
    if (whole) return *reinterpret_cast<const uint32_t*>(S + (ro + (unsigned)gx0));
    if (gx0 >= w + 3) return 0u;   // a dword wholly right of the reflected fringe (columns w .. w + 2) feeds no stored output: no byte loads for it
    uint32_t v = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) v |= (uint32_t)S[ro + (unsigned)reflect101(gx0 + j, w)] << (8 * j);
    return v;
}

// The same dword WITHOUT byte loads (4-byte aligned planes): every staged dword, also one that straddles or lies beyond an image
// edge, is a v_perm of two aligned dwords of the row -- which two and with which selector depends only on the column
// (REFLECT_101 of columns -4 .. -1 and w .. w + 2; further out nothing is needed), so a thread works it out once and
// every lane of every wave runs the same three instructions per fetch.  (The byte path made EVERY wave of an edge strip
// run 4 byte loads + reflections per dword: edge strips cost 2.4 x an inner one.)
struct BlurCol { unsigned a0, a1, sel; };
__device__ __forceinline__ BlurCol blur_col(int g, int w, unsigned ss) {
#define BLUR_SEL(i0, i1, i2, i3) ((unsigned)(i0) | ((unsigned)(i1) << 8) | ((unsigned)(i2) << 16) | ((unsigned)(i3) << 24))
    BlurCol c;
    if (g < 0) { c.a0 = 0u; c.a1 = 4u; c.sel = BLUR_SEL(4, 3, 2, 1); }                       // columns -4 .. -1 = 4, 3, 2, 1
    else if (g + 4 <= w) { c.a0 = c.a1 = (unsigned)g; c.sel = BLUR_SEL(0, 1, 2, 3); }
    else if (g < w) {                                                                         // q real columns, then w - 2, w - 3, ..
        const int q = w - g;
        unsigned idx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) idx[j] = (unsigned)(j < q ? 4 + j : 2 * q + 2 - j);
        c.a0 = (unsigned)(g - 4); c.a1 = (unsigned)g; c.sel = BLUR_SEL(idx[0], idx[1], idx[2], idx[3]);
    } else if (g <= w + 2) {                                                                  // columns w + t + j = w - 2 - t - j, needed up to w + 2
        const int t = g - w, gb = (w - 4) & ~3;
        unsigned idx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) idx[j] = j <= 2 - t ? (unsigned)(w - 2 - t - j - gb) : 0x0Cu;
        c.a0 = (unsigned)gb; c.a1 = min((unsigned)gb + 4u, ss - 4u); c.sel = BLUR_SEL(idx[0], idx[1], idx[2], idx[3]);
    } else { c.a0 = c.a1 = 0u; c.sel = BLUR_SEL(0x0C, 0x0C, 0x0C, 0x0C); }
#undef BLUR_SEL
    return c;
}
__device__ __forceinline__ uint32_t blur_fetch2(const uint8_t* S, unsigned ss, int h, int row, const BlurCol c) {
    const unsigned ro = __umul24((unsigned)reflect101(row, h), ss);
    return __builtin_amdgcn_perm(*reinterpret_cast<const uint32_t*>(S + (ro + c.a1)), *reinterpret_cast<const uint32_t*>(S + (ro + c.a0)), c.sel);
}

template <bool INTERIOR>   // INTERIOR: the strip and its halo lie inside the level and the plane is 4-byte aligned -- no reflection, no byte path, no predicates
__device__ __forceinline__ void blur_strip(uint8_t (*in)[72], uint32_t (*rp)[64], const uint8_t* S, const size_t sstride, const bool aligned,
                                           uint8_t* D, const unsigned dstride, const int w, const int h, const int bx, const int by0, const int tid) {
    // in[r][k] holds column bx - 4 + k of row by - 3 + r; a thread stages dwords i0 and i1 of the 22 x 18 of a step
    const int i0 = tid, i1 = tid + 256;
    const bool has1 = i1 < 22 * 18;
    const int r0 = i0 / 18, r1 = i1 / 18;
    const int g0 = bx - 4 + 4 * (i0 - r0 * 18), g1 = bx - 4 + 4 * (i1 - r1 * 18);
    const bool whole0 = aligned && g0 >= 0 && g0 + 4 <= w, whole1 = aligned && g1 >= 0 && g1 + 4 <= w;
    uint32_t* const st0 = reinterpret_cast<uint32_t*>(&in[r0][4 * (i0 - r0 * 18)]);
    uint32_t* const st1 = reinterpret_cast<uint32_t*>(&in[has1 ? r1 : 0][has1 ? 4 * (i1 - r1 * 18) : 0]);
    const unsigned ss = (unsigned)sstride;
    const BlurCol bc0 = blur_col(g0, w, ss), bc1 = blur_col(g1, w, ss);   // (used by edge strips of aligned planes)
    uint32_t v0 = INTERIOR ? *reinterpret_cast<const uint32_t*>(S + (__umul24((unsigned)(by0 + r0 - 3), ss) + (unsigned)g0))
                           : aligned ? blur_fetch2(S, ss, h, by0 + r0 - 3, bc0) : blur_fetch(S, ss, whole0, w, h, g0, by0 + r0 - 3);
    uint32_t v1 = !has1 ? 0u : INTERIOR ? *reinterpret_cast<const uint32_t*>(S + (__umul24((unsigned)(by0 + r1 - 3), ss) + (unsigned)g1))
                                        : aligned ? blur_fetch2(S, ss, h, by0 + r1 - 3, bc1) : blur_fetch(S, ss, whole1, w, h, g1, by0 + r1 - 3);
    *st0 = v0;
    if (has1) *st1 = v1;
    __syncthreads();
    // taps as dot-product operands: bytes for the row pass, halfword pairs for the column pass (which pair of
    // rows a tap pair meets depends on the parity of the output row)
    const uint32_t TA = 18u | (34u << 8) | (49u << 16) | (55u << 24), TB = 49u | (34u << 8) | (18u << 16);
    const int cy = tid >> 4, cx4 = (tid & 15) * 4;
    const bool odd = cy & 1;
    const uint32_t k0 = odd ? (18u << 16) : (18u | (34u << 16)), k1 = odd ? (34u | (49u << 16)) : (49u | (55u << 16));
    const uint32_t k2 = odd ? (55u | (49u << 16)) : (49u | (34u << 16)), k3 = odd ? (34u | (18u << 16)) : 18u;
#pragma unroll 1
    for (int s = 0; s < BLUR_STEPS; ++s) {
        const int by = by0 + 16 * s;
        const bool more = s + 1 < BLUR_STEPS && (INTERIOR || by + 16 < h);
        if (more) {
            if (INTERIOR) {
                v0 = *reinterpret_cast<const uint32_t*>(S + (__umul24((unsigned)(by + 16 + r0 - 3), ss) + (unsigned)g0));
                if (has1) v1 = *reinterpret_cast<const uint32_t*>(S + (__umul24((unsigned)(by + 16 + r1 - 3), ss) + (unsigned)g1));
            } else {
                if (aligned) {
                    v0 = blur_fetch2(S, ss, h, by + 16 + r0 - 3, bc0);
                    if (has1) v1 = blur_fetch2(S, ss, h, by + 16 + r1 - 3, bc1);
                } else {
                    v0 = blur_fetch(S, ss, whole0, w, h, g0, by + 16 + r0 - 3);
                    if (has1) v1 = blur_fetch(S, ss, whole1, w, h, g1, by + 16 + r1 - 3);
                }
            }
        }
        // row pass: thread -> 4 adjacent outputs of TWO rows (2p, 2p+1).  Output o needs the 7 bytes at columns
        // c+o-3 .. c+o+3 = stream bytes o+1 .. o+7 of three ALIGNED dwords (columns c-4 .. c+7): two v_dot4_u32_u8
        // on byte quads cut out with v_alignbyte.  (Unaligned LDS reads cost ~30 cycles each on gfx950.)
        if (tid < 11 * 16) {
            const int p = tid >> 4, c = (tid & 15) * 4;
            uint32_t o[2][4];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const uint32_t* pw = reinterpret_cast<const uint32_t*>(&in[2 * p + rr][c]);
                const uint32_t w0 = pw[0], w1 = pw[1], w2 = pw[2];
                o[rr][0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 1), TB, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 1), TA, 0u, false), false);
                o[rr][1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 2), TB, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 2), TA, 0u, false), false);
                o[rr][2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w2, w1, 3), TB, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(w1, w0, 3), TA, 0u, false), false);
                o[rr][3] = __builtin_amdgcn_udot4(w2, TB, __builtin_amdgcn_udot4(w1, TA, 0u, false), false);
            }
            *reinterpret_cast<uint4*>(&rp[p][c]) = make_uint4(o[0][0] | (o[1][0] << 16), o[0][1] | (o[1][1] << 16),
                                                               o[0][2] | (o[1][2] << 16), o[0][3] | (o[1][3] << 16));
        }
        __syncthreads();
        if (more) {
            *st0 = v0;
            if (has1) *st1 = v1;
        }
        if (INTERIOR || (bx + cx4 < w && by + cy < h)) {
            // output row cy uses rows cy .. cy+6: four row pairs starting at pair cy >> 1
            const int pb = cy >> 1;
            const uint4 a0 = *reinterpret_cast<const uint4*>(&rp[pb][cx4]), a1 = *reinterpret_cast<const uint4*>(&rp[pb + 1][cx4]);
            const uint4 a2 = *reinterpret_cast<const uint4*>(&rp[pb + 2][cx4]), a3 = *reinterpret_cast<const uint4*>(&rp[pb + 3][cx4]);
#define BLUR_COL(f) min((udot2(a3.f, k3, udot2(a2.f, k2, udot2(a1.f, k1, udot2(a0.f, k0, 1u << 15)))) >> 16), 255u)
            const uint32_t out = BLUR_COL(x) | (BLUR_COL(y) << 8) | (BLUR_COL(z) << 16) | (BLUR_COL(w) << 24);
#undef BLUR_COL
            *reinterpret_cast<uint32_t*>(D + (__umul24((unsigned)(by + cy), dstride) + (unsigned)(bx + cx4))) = out;
        }
        if (!more) break;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void blur_all_kernel(
    const OrbLevel* __restrict__ levels, const uint4* __restrict__ tiles,
    const uint8_t* __restrict__ img0, size_t img0_stride, size_t img0_frame,
    const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, unsigned ntiles, unsigned inv_tiles, unsigned total) {
    __shared__ __attribute__((aligned(16))) uint8_t in[22][72];
    // row-pass results as VERTICAL PAIRS: rp[p][c] = row 2p | row 2p+1 << 16 (each <= 255 * 257), the operand
    // shape of v_dot2_u32_u16 in the column pass
    __shared__ __attribute__((aligned(16))) uint32_t rp[11][64];
    const int tid = threadIdx.x;
    // XCD-aware order: workgroup i runs on XCD i % 8, so XCD k takes the k-th contiguous eighth of the (frame, strip)
    // list -- strips that share halo columns and rows then share one L2 instead of fetching them into eight
    const unsigned lin = blockIdx.x, per_xcd = (total + 7u) >> 3;
    const unsigned logical = (lin & 7u) * per_xcd + (lin >> 3);
    if (logical >= total) return;
    int frame = (int)__umulhi(logical, inv_tiles);
    if ((unsigned)frame * ntiles > logical) --frame;   // the rounded-up reciprocal can overshoot by one on huge grids
    const uint4 tt = tiles[logical - (unsigned)frame * ntiles];   // host-built (orbk_blur_tiles): level, strip origin -- no level search
    const int level = (int)tt.x, bx = (int)tt.y, by0 = (int)tt.z;
    const OrbLevel& L = levels[level];
    const int w = L.w, h = L.h;
    const uint8_t* S;
    size_t sstride;
    if (level == 0) { S = img0 + (size_t)frame * img0_frame; sstride = img0_stride; }
    else { S = pyr + L.plane_off + (size_t)frame * L.plane_bytes; sstride = (size_t)L.stride; }
    const bool aligned = ((((uintptr_t)S) | sstride) & 3) == 0;
    uint8_t* D = blur + L.blur_off + (size_t)frame * L.blur_bytes;
    // most strips lie inside their level: they take the copy of the loop without reflection, byte path and predicates
    const bool interior = aligned && bx >= 4 && bx + 68 <= w && by0 >= 3 && by0 + 16 * BLUR_STEPS + 3 <= h;
    if (interior) blur_strip<true>(in, rp, S, sstride, aligned, D, (unsigned)L.stride, w, h, bx, by0, tid);
    else blur_strip<false>(in, rp, S, sstride, aligned, D, (unsigned)L.stride, w, h, bx, by0, tid);
}

// K6, streaming form, for the strips whose COLUMNS lie inside the level (dwords bx - 4 .. bx + 67 inside the row, planes
// 4-byte aligned: no byte path, no column reflection -- three quarters of the pixels).  One THREAD walks a 4-pixel column
// group down the strip's 64 rows: per pair of input rows one 12-byte load each, the row pass (14 operations per row, as
// above), the two rows packed as the halfwords of one dword per column; the last four such pairs are the whole state
// (16 registers), and every new pair completes TWO output rows (y = 2p - 3 and 2p - 2 for the pair of rows 2p, 2p + 1,
// with the two tap patterns of v_dot2_u32_u16 above).  No LDS, no barrier, no halo rows staged twice: ~0.2 instructions
// per pixel and wave against 0.37 (0.66 with the byte path) of the tile form.  Rows reflect by index (REFLECT_101).
// A workgroup of 256 threads takes 16 strips.  The strips at the left / right image edge take the EDGE instance (their own launch):
// the same walk with each window dword built from two aligned dwords (blur_col), and no thread for columns past the image.
template <bool EDGE>   // EDGE: strips that touch the left / right image edge -- each of the three window dwords is a v_perm of two aligned dwords (blur_col)
__global__ __launch_bounds__(256) void blur_stream_kernel(
    const OrbLevel* __restrict__ levels, const uint4* __restrict__ tiles,
    const uint8_t* __restrict__ img0, size_t img0_stride, size_t img0_frame,
    const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, unsigned ntiles) {
    const unsigned tile = blockIdx.x * 16u + (threadIdx.x >> 4);
    if (tile >= ntiles) return;
    const int frame = blockIdx.y;
    const uint4 tt = tiles[tile];
    const int level = (int)tt.x, by0 = (int)tt.z;
    const int x0 = (int)tt.y + 4 * (int)(threadIdx.x & 15);
    const OrbLevel& L = levels[level];
    const int h = L.h;
    const uint8_t* S;
    unsigned ss;
    if (level == 0) { S = img0 + (size_t)frame * img0_frame; ss = (unsigned)img0_stride; }
    else { S = pyr + L.plane_off + (size_t)frame * L.plane_bytes; ss = (unsigned)L.stride; }
    if (EDGE && x0 >= L.w) return;   // (a strip at the right edge is rarely 64 columns wide)
    BlurCol bc[3];
    if (EDGE) {
#pragma unroll
        for (int k = 0; k < 3; ++k) bc[k] = blur_col(x0 - 4 + 4 * k, L.w, ss);
    } else S += x0 - 4;
    uint8_t* D = blur + L.blur_off + (size_t)frame * L.blur_bytes + x0;
    const unsigned ds = (unsigned)L.stride;
    const uint32_t TA = 18u | (34u << 8) | (49u << 16) | (55u << 24), TB = 49u | (34u << 8) | (18u << 16);
    const uint32_t kA0 = 18u | (34u << 16), kA1 = 49u | (55u << 16), kA2 = 49u | (34u << 16), kA3 = 18u;           // output row 2p - 3
    const uint32_t kB0 = 18u << 16, kB1 = 34u | (49u << 16), kB2 = 55u | (49u << 16), kB3 = 34u | (18u << 16);     // output row 2p - 2
    const int ylast = min(by0 + 16 * BLUR_STEPS, h) - 1;
    const int p0 = (by0 - 3) >> 1, p1 = (ylast + 3) >> 1;   // pairs of rows (2p, 2p + 1); arithmetic shift: by0 - 3 may be negative
    typedef unsigned blur_u3 __attribute__((ext_vector_type(3)));
    auto fetch = [&](int yy) -> blur_u3 {
        int r = yy < 0 ? -yy : yy;
        r = r >= h ? 2 * h - 2 - r : r;
        r = min(max(r, 0), h - 1);   // (rows further out are only read for outputs that are not stored)
        if (EDGE) {
            const uint8_t* R = S + __umul24((unsigned)r, ss);
            blur_u3 v;
            v.x = __builtin_amdgcn_perm(*reinterpret_cast<const uint32_t*>(R + bc[0].a1), *reinterpret_cast<const uint32_t*>(R + bc[0].a0), bc[0].sel);
            v.y = __builtin_amdgcn_perm(*reinterpret_cast<const uint32_t*>(R + bc[1].a1), *reinterpret_cast<const uint32_t*>(R + bc[1].a0), bc[1].sel);
            v.z = __builtin_amdgcn_perm(*reinterpret_cast<const uint32_t*>(R + bc[2].a1), *reinterpret_cast<const uint32_t*>(R + bc[2].a0), bc[2].sel);
            return v;
        }
        return *reinterpret_cast<const blur_u3*>(S + __umul24((unsigned)r, ss));   // rows < 2^16, pitch < 2^24: the full-rate multiply
    };
    auto rowpass = [&](const blur_u3 v, uint32_t* o) {
        o[0] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(v.z, v.y, 1), TB, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(v.y, v.x, 1), TA, 0u, false), false);
        o[1] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(v.z, v.y, 2), TB, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(v.y, v.x, 2), TA, 0u, false), false);
        o[2] = __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(v.z, v.y, 3), TB, __builtin_amdgcn_udot4(__builtin_amdgcn_alignbyte(v.y, v.x, 3), TA, 0u, false), false);
        o[3] = __builtin_amdgcn_udot4(v.z, TB, __builtin_amdgcn_udot4(v.y, TA, 0u, false), false);
    };
    uint32_t P0[4] = {0, 0, 0, 0}, P1[4] = {0, 0, 0, 0}, P2[4] = {0, 0, 0, 0}, P3[4];   // the last four row pairs, oldest first
    // three pairs of rows in flight: a thread's walk is a chain of ~36 dependent steps, and with one pair ahead every step waited for HBM
    blur_u3 va = fetch(2 * p0), vb = fetch(2 * p0 + 1), va1 = fetch(2 * p0 + 2), vb1 = fetch(2 * p0 + 3), va2 = fetch(2 * p0 + 4), vb2 = fetch(2 * p0 + 5);
#pragma unroll 4
    for (int p = p0; p <= p1; ++p) {
        const blur_u3 ca = va, cb = vb;
        va = va1; vb = vb1; va1 = va2; vb1 = vb2;
        va2 = fetch(2 * p + 6); vb2 = fetch(2 * p + 7);   // (rows past the strip's last pair are clamped reads of rows that exist)
        uint32_t oa[4], ob[4];
        rowpass(ca, oa); rowpass(cb, ob);
#pragma unroll
        for (int j = 0; j < 4; ++j) P3[j] = oa[j] | (ob[j] << 16);
        const int ya = 2 * p - 3, yb = 2 * p - 2;
#define BLUR_OUT(k0, k1, k2, k3, j) min((udot2(P3[j], k3, udot2(P2[j], k2, udot2(P1[j], k1, udot2(P0[j], k0, 1u << 15)))) >> 16), 255u)
        if (ya >= by0 && ya <= ylast)
            *reinterpret_cast<uint32_t*>(D + __umul24((unsigned)ya, ds)) = BLUR_OUT(kA0, kA1, kA2, kA3, 0) | (BLUR_OUT(kA0, kA1, kA2, kA3, 1) << 8) | (BLUR_OUT(kA0, kA1, kA2, kA3, 2) << 16) | (BLUR_OUT(kA0, kA1, kA2, kA3, 3) << 24);
        if (yb >= by0 && yb <= ylast)
            *reinterpret_cast<uint32_t*>(D + __umul24((unsigned)yb, ds)) = BLUR_OUT(kB0, kB1, kB2, kB3, 0) | (BLUR_OUT(kB0, kB1, kB2, kB3, 1) << 8) | (BLUR_OUT(kB0, kB1, kB2, kB3, 2) << 16) | (BLUR_OUT(kB0, kB1, kB2, kB3, 3) << 24);
#undef BLUR_OUT
#pragma unroll
        for (int j = 0; j < 4; ++j) { P0[j] = P1[j]; P1[j] = P2[j]; P2[j] = P3[j]; }
    }
}

// --------------------------------------------------------------------------------------------
// K7: rotated BRIEF (256 tests) + output assembly.  One wavefront per keypoint: lane l evaluates
// tests 4l..4l+3 (a nibble), lane pairs combine to one descriptor byte.  The wave also writes the
// final cv::KeyPoint-shaped record at its level-major position and lane 0 of keypoint 0 of
// level 0 writes the frame total.
// --------------------------------------------------------------------------------------------
__constant__ __attribute__((aligned(16))) signed char c_pattern[SLAMIT_ORB_PATTERN_INTS];
#define DESC_KP_PER_WAVE 4
#define DESC_R 18                 // reach of the rotated pattern
#define DESC_ROWS (2 * DESC_R + 1)
#define DESC_PITCH 48             // 37 columns + up to 3 of alignment, in 16-byte chunks
#define DESC_LOADS 2              // ceil(37 * 3 / 64) 16-byte chunks per lane and keypoint

__global__ __launch_bounds__(256) void describe_kernel(
    const OrbLevel* __restrict__ levels, int nlevels,
    const uint8_t* __restrict__ blur,
    const OrbLevelKp* __restrict__ lkp, size_t kp_frame_stride, const int* __restrict__ kp_count,
    slamit_kp* __restrict__ out_kps, uint8_t* __restrict__ out_desc, int out_cap,
    int* __restrict__ out_n, int xcd_frames /* 0: grid (blocks, levels, frames); else the frame count of a 1-D grid */, int kblocks) {
    const int lane = threadIdx.x & 63;
    int level, frame, kb;
    if (xcd_frames) {
        // 1-D grid, frames dealt to the XCDs: consecutive workgroup ids go to consecutive XCDs (an L2 each), so workgroup w works on frame
        // 8 (k / P) + (w & 7) with k = w >> 3 and P workgroups per frame -- an XCD walks ITS frames one after the other and the patches of
        // a frame's neighbouring keypoints meet in one L2 instead of eight
        const unsigned w = blockIdx.x, k = w >> 3, P = (unsigned)(nlevels * kblocks), fq = k / P, local = k - fq * P;
        frame = (int)(8u * fq + ((w + fq) & 7u));   // (rotated per group of eight frames: an XCD does not keep meeting every eighth frame of a periodic input)
        if (frame >= xcd_frames) return;
        level = (int)(local / (unsigned)kblocks); kb = (int)(local - (unsigned)level * (unsigned)kblocks);
    } else { level = blockIdx.y; frame = blockIdx.z; kb = blockIdx.x; }
    const int i0 = (kb * 4 + (threadIdx.x >> 6)) * DESC_KP_PER_WAVE;
    const OrbLevel& L = levels[level];
    const int* counts = kp_count + frame * nlevels;
    int offset = 0, total = 0;
    for (int l = 0; l < nlevels; ++l) {
        int c = counts[l];
        if (l < level) offset += c;
        total += c;
    }
    if (level == 0 && i0 == 0 && lane == 0) out_n[frame] = min(total, out_cap);
    const int count = counts[level];
    if (i0 >= count) return;
    // the lane's four tests (8 pattern points) are the same for every keypoint: loaded once per wave, and the
    // keypoints of a wave are processed with all their loads in flight together (the kernel is gather-latency bound)
    const int4 pw = reinterpret_cast<const int4*>(c_pattern)[lane];
    const int pt[4] = {pw.x, pw.y, pw.z, pw.w};
    float px0[4], py0[4], px1[4], py1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        px0[t] = (float)(signed char)(pt[t] & 0xFF); py0[t] = (float)(signed char)((pt[t] >> 8) & 0xFF);
        px1[t] = (float)(signed char)((pt[t] >> 16) & 0xFF); py1[t] = (float)(signed char)((pt[t] >> 24) & 0xFF);
    }
    const uint8_t* img = blur + L.blur_off + (size_t)frame * L.blur_bytes;
    const unsigned step = (unsigned)L.stride;
    const OrbLevelKp* KP = lkp + L.kp_off + (size_t)frame * kp_frame_stride;
    OrbLevelKp kp[DESC_KP_PER_WAVE];
#pragma unroll
    for (int k = 0; k < DESC_KP_PER_WAVE; ++k) kp[k] = KP[min(i0 + k, count - 1)];
    // The rotated pattern reaches 18 px from the keypoint (max |point| = 18.4): the 37 x 37 patch is staged in LDS
    // with row-contiguous aligned dword loads (a byte gather straight from the plane touches a different cache
    // line per row and is bound by the L1 tag rate), then the 8 taps per lane are LDS byte reads.
    __shared__ __attribute__((aligned(16))) uint8_t s_patch[4][DESC_KP_PER_WAVE][DESC_ROWS * DESC_PITCH];
    const int wv = threadIdx.x >> 6;
    // a lane stages chunk (row, c) = (idx / 3, idx % 3) of a patch for idx = lane and lane + 64: ONE 16-byte buffer load each (2 per
    // keypoint instead of 7 dword loads: the pass waits on vector-memory issue like the pyramid did), bounded by the plane's
    // buffer resource (a chunk may run up to 10 bytes past a row end: the next row, or zeros past the plane)
    int lrow[DESC_LOADS], lcol[DESC_LOADS];
    unsigned poff[DESC_LOADS];
#pragma unroll
    for (int it = 0; it < DESC_LOADS; ++it) {
        const int idx = lane + 64 * it;
        lrow[it] = (int)(((unsigned)idx * 21846u) >> 16);   // idx / 3 for idx < 128
        lcol[it] = idx - 3 * lrow[it];
        poff[it] = __umul24((unsigned)min(lrow[it], DESC_ROWS - 1), step) + 16u * (unsigned)lcol[it];
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(img), 0, (unsigned)L.stride * (unsigned)L.h, 0x00020000);
    typedef unsigned desc_u4 __attribute__((ext_vector_type(4)));
    desc_u4 ld[DESC_KP_PER_WAVE][DESC_LOADS];
#pragma unroll
    for (int k = 0; k < DESC_KP_PER_WAVE; ++k) {
        const unsigned base = __umul24((unsigned)(kp[k].y - DESC_R), step) + (unsigned)((kp[k].x - DESC_R) & ~3);
#pragma unroll
        for (int it = 0; it < DESC_LOADS; ++it) ld[k][it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(base + poff[it]), 0, 0);
    }
#pragma unroll
    for (int k = 0; k < DESC_KP_PER_WAVE; ++k)
#pragma unroll
        for (int it = 0; it < DESC_LOADS; ++it)
            if (lrow[it] < DESC_ROWS) *reinterpret_cast<desc_u4*>(&s_patch[wv][k][__mul24(lrow[it], DESC_PITCH) + 16 * lcol[it]]) = ld[k][it];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // The rotated tap (row, col) = (cvRound(x b + y a), cvRound(x a - y b)) of a pattern point (ORBextractor.cc:121-123) on packed
    // f32 lanes: x * (b, a) + y * (a, -b) is two v_pk_mul_f32 and one v_pk_add_f32 with the same four products and two sums
    // (x a - y b == x a + (-(y b)) bit for bit, no contraction); adding 1.5 * 2^23 rounds both halves to the nearest integer,
    // ties to even like cvRound, and leaves it in the low mantissa bits (|value| <= 18.4 << 2^22), so the LDS address
    // row * PITCH + col is one 24-bit multiply-add of the two bit patterns plus a per-keypoint base that carries the
    // constant -- 6 vector instructions per tap instead of 13 (4 multiplies, 2 adds, 2 x (v_rndne + v_cvt), address).
    typedef float desc_f2 __attribute__((ext_vector_type(2)));
    const desc_f2 magic = {12582912.0f, 12582912.0f};
    // low 24 bits of (0x4B400000 + r) = 0x400000 + r, so umul24(bits_r, PITCH) + bits_c = r PITCH + c + DESC_BIAS
    const unsigned DESC_BIAS = 0x400000u * (unsigned)DESC_PITCH + 0x4B400000u;
    int tv0[DESC_KP_PER_WAVE][4], tv1[DESC_KP_PER_WAVE][4];
#pragma unroll
    for (int k = 0; k < DESC_KP_PER_WAVE; ++k) {
        const float a = kp[k].cs, b = kp[k].sn;  // a = cos, b = sin   (ORBextractor.cc:117-118; computed by ic_angle_kernel)
        const desc_f2 ba = {b, a}, anb = {a, -b};
        const uint8_t* center = &s_patch[wv][k][DESC_R * DESC_PITCH + DESC_R + ((kp[k].x - DESC_R) & 3)];
        const uint8_t* biased = center - DESC_BIAS;   // (an LDS address: 32-bit wrap-around arithmetic)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const desc_f2 p0 = {px0[t], px0[t]}, q0 = {py0[t], py0[t]}, p1 = {px1[t], px1[t]}, q1 = {py1[t], py1[t]};
            const desc_f2 m0 = (p0 * ba + q0 * anb) + magic, m1 = (p1 * ba + q1 * anb) + magic;
            tv0[k][t] = biased[__umul24(__float_as_uint(m0.x), (unsigned)DESC_PITCH) + __float_as_uint(m0.y)];
            tv1[k][t] = biased[__umul24(__float_as_uint(m1.x), (unsigned)DESC_PITCH) + __float_as_uint(m1.y)];
        }
    }
#pragma unroll
    for (int k = 0; k < DESC_KP_PER_WAVE; ++k) {
        const int i = i0 + k, o = offset + i;
        unsigned nib = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t) nib |= (unsigned)(tv0[k][t] < tv1[k][t]) << t;
        unsigned other = __shfl_xor(nib, 1, WAVE);
        if (i >= count || o >= out_cap) continue;
        if ((lane & 1) == 0)
            out_desc[((size_t)frame * out_cap + o) * SLAMIT_DESC_BYTES + (lane >> 1)] = (uint8_t)(nib | (other << 4));
        if (lane == 0) {
            slamit_kp kk;
            float fx = (float)kp[k].x, fy = (float)kp[k].y;
            if (level != 0) { fx *= L.scale; fy *= L.scale; }  // ORBextractor.cc:1126-1132
            kk.x = fx; kk.y = fy; kk.size = L.patch_size; kk.angle = kp[k].angle; kk.response = kp[k].response;
            kk.octave = level; kk.class_id = -1;
            out_kps[(size_t)frame * out_cap + o] = kk;
        }
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
    static const DiscTable disc = make_disc();
    hipError_t e = hipMemcpyToSymbolAsync(HIP_SYMBOL(c_disc), &disc, sizeof(disc), 0, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
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

hipError_t orbk_pyramid_prepare(int smem_bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(pyramid_fused_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes);
}

void orbk_pyramid(hipStream_t st, const OrbLevel* levels, int nlevels, const PyrBox* boxes, const PyrTabs* tabs,
                  int nregions, const uint8_t* img0, size_t img0_stride, size_t img0_frame, uint8_t* pyr, int bufA_bytes,
                  int smem_bytes, int nframes, int l_first, int l_last, int threads) {
    hipLaunchKernelGGL(pyramid_fused_kernel, dim3(nregions, nframes), dim3(threads), smem_bytes, st, levels, nlevels, boxes,
                       tabs, img0, img0_stride, img0_frame, pyr, bufA_bytes, l_first, l_last);
}

#ifndef FAST_PS
#define FAST_PS 48   // tile pitch for cells of <= 37 pixels
#define FAST_PL 80   // ... up to 64 (a multiple of 16: the tile is staged in 16-byte DMA chunks)
#endif
static int fast_pitch(int max_wcell) { return max_wcell + 11 <= FAST_PS ? FAST_PS : FAST_PL; }  // LDS columns 0 .. sw + 10 are touched

size_t orbk_fast_smem(int max_wcell, int max_hcell) {
    const int P = fast_pitch(max_wcell);
    const int tile_rows = max_hcell + 6, sc_rows = max_hcell + 2;
    const int kp_cap = ((max_wcell + 1) / 2) * ((max_hcell + 1) / 2);
    return (size_t)4 * fast_wave_bytes(P, tile_rows, sc_rows, kp_cap);
}

hipError_t orbk_fast_prepare(int max_wcell, int max_hcell) {
    const int smem = (int)orbk_fast_smem(max_wcell, max_hcell);
    hipError_t e;
    if (fast_pitch(max_wcell) == FAST_PS) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(fast_cells_kernel<FAST_PS>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(fast_cells_kernel<FAST_PL>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    }
    return e;
}

// Cell table: the non-empty FAST cells of every level in the reference's visiting order (level, row, column;
// ORBextractor.cc:805-822), 4 words per cell:
//   0: level | cell index in the level << 8      1: iniX | iniY << 16      2: cw | ch << 8 (window size)
//   3: division magic  floor(2^20 / g) + 1  for g = (sw + 3) / 4, the groups of four scan pixels per row
int orbk_fast_cells(const OrbLevel* host_levels, int nlevels, std::vector<uint32_t>& out) {
    out.clear();
    auto magic = [](int g) { return (uint32_t)((1u << 20) / (unsigned)std::max(g, 1) + 1u); };
    for (int l = 0; l < nlevels; ++l) {
        const OrbLevel& L = host_levels[l];
        for (int c = 0; c < L.ncells; ++c) {
            const int ci = c / L.nCols, cj = c - ci * L.nCols;
            const int iniX = ORB_MIN_BORDER + cj * L.wCell, iniY = ORB_MIN_BORDER + ci * L.hCell;
            if (iniY >= L.maxBorderY - 3 || iniX >= L.maxBorderX - 6) continue;  // ORBextractor.cc:810,819
            const int cw = std::min(L.wCell + 6, L.maxBorderX - iniX), ch = std::min(L.hCell + 6, L.maxBorderY - iniY);
            const int sw = cw - 6, sh = ch - 6;
            if (sw <= 0 || sh <= 0) continue;
            const uint32_t w[4] = {(uint32_t)l | ((uint32_t)c << 8), (uint32_t)iniX | ((uint32_t)iniY << 16),
                                   (uint32_t)cw | ((uint32_t)ch << 8), magic((sw + 3) >> 2)};
            out.insert(out.end(), w, w + 4);
        }
    }
    return (int)(out.size() / 4);
}

void orbk_fast(hipStream_t st, const OrbLevel* host_levels, int nlevels, const uint32_t* d_cells, int cells_per_frame,
               const uint8_t* img0, size_t img0_stride, size_t img0_frame, const uint8_t* pyr,
               unsigned long long* cand, size_t cand_frame_stride, int* cand_count, int iniTh, int minTh,
               int max_wcell, int max_hcell, int nframes) {
    const int tile_rows = max_hcell + 6, sc_rows = max_hcell + 2;
    const int kp_cap = ((max_wcell + 1) / 2) * ((max_hcell + 1) / 2);
    // The waves of a workgroup need nothing from each other, and a four-wave workgroup holds its LDS until its slowest cell is done
    // (cells differ 3x in work).  One wave (= one cell) per workgroup, cells dealt to the XCDs in runs of four (fast_cells_kernel):
    // 0.283 -> 0.278 ms per 256 VGA frames at the same HBM traffic; at 1280 x 720 (2,656 cells per frame) four per workgroup stay
    // 1 - 2 % ahead at 16 - 128 frames, so the choice goes by the frame's cell count -- two measured geometries, no model.
    // (SLAMIT_FAST_WPB=1|2|4: A/B runs.)
    static const int wpb_env = getenv("SLAMIT_FAST_WPB") ? atoi(getenv("SLAMIT_FAST_WPB")) : 0;
    const int wpb = wpb_env == 1 || wpb_env == 2 || wpb_env == 4 ? wpb_env : cells_per_frame < 1600 ? 1 : 4;
    static const int xcd_env = getenv("SLAMIT_XCD_FRAMES") ? atoi(getenv("SLAMIT_XCD_FRAMES")) : 1;
    const int xcd_frames = xcd_env && nframes >= 16 ? nframes : 0;
    const dim3 grid = xcd_frames ? dim3((unsigned)(((nframes + 7) / 8) * 8) * (unsigned)((cells_per_frame + wpb - 1) / wpb))
                                 : dim3(wpb == 1 ? (unsigned)((cells_per_frame + 31) & ~31) : (unsigned)((cells_per_frame + wpb - 1) / wpb), nframes);
    const size_t smem = orbk_fast_smem(max_wcell, max_hcell) / 4 * wpb;
    FastTab tab = {};
    for (int l = 0; l < nlevels && l < ORB_MAX_LEVELS; ++l) {
        const OrbLevel& S = host_levels[l];
        FastLevel& D = tab.lv[l];
        D.cell_base = S.cell_base; D.nCols = S.nCols; D.wCell = S.wCell; D.hCell = S.hCell;
        D.maxBorderX = S.maxBorderX; D.maxBorderY = S.maxBorderY; D.stride = S.stride; D.cand_cap = S.cand_cap;
        D.plane_off = S.plane_off; D.plane_bytes = S.plane_bytes; D.cand_off = S.cand_off;
    }
    if (fast_pitch(max_wcell) == FAST_PS)
        hipLaunchKernelGGL(fast_cells_kernel<FAST_PS>, grid, dim3(64 * wpb), smem, st, tab, reinterpret_cast<const uint4*>(d_cells), nlevels, cells_per_frame, img0,
                           (unsigned)img0_stride, img0_frame, pyr, cand, cand_frame_stride, cand_count, iniTh, minTh, tile_rows,
                           sc_rows, kp_cap, xcd_frames);
    else
        hipLaunchKernelGGL(fast_cells_kernel<FAST_PL>, grid, dim3(64 * wpb), smem, st, tab, reinterpret_cast<const uint4*>(d_cells), nlevels, cells_per_frame, img0,
                           (unsigned)img0_stride, img0_frame, pyr, cand, cand_frame_stride, cand_count, iniTh, minTh, tile_rows,
                           sc_rows, kp_cap, xcd_frames);
}

// candidates kept in LDS: as many as fit beside the node arrays in half a CU's LDS (lists above that use the HBM workspace)
// LDS budget of one octree workgroup.  Two 78 KB workgroups fill a CU's LDS, which also keeps every other kernel off
// the chip while the octree pass (a few hundred workgroups, latency bound) runs; images up to about VGA rarely have more
// than 5,000 candidates on a level, so their handles take 48 KB and the blur runs beside the octree on the side stream.
int orbk_octree_key_cap(int node_cap, int width, int height) {
    long budget = (long)width * height <= 640L * 480L * 3 / 2 ? 48L * 1024 : (long)OCT_LDS_BUDGET;
    if (getenv("SLAMIT_OCT_LDS_KB")) budget = 1024L * atol(getenv("SLAMIT_OCT_LDS_KB"));
    const long room = budget - (long)orbk_octree_node_bytes(node_cap);
    if (getenv("SLAMIT_OCT_KEYS")) return atoi(getenv("SLAMIT_OCT_KEYS")) & ~7;
    return (int)std::min<long>(OCT_LDS_KEYS_MAX, std::max<long>(OCT_LDS_KEYS_MIN, room / 6)) & ~7;
}
size_t orbk_octree_smem(int node_cap, int key_cap) { return orbk_octree_node_bytes(node_cap) + (size_t)key_cap * 6; }

hipError_t orbk_octree_prepare(int node_cap, int key_cap) {
    static int prepared = 0;   // the attribute is per function, not per handle: keep the largest request
    const int want = (int)orbk_octree_smem(node_cap, key_cap);
    if (want <= prepared) return hipSuccess;
    prepared = want;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(octree_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize, want);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(octree_kernel<OCT_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, want);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(octree_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, want);
}

void orbk_octree(hipStream_t st, const OrbLevel* levels, int nlevels, const unsigned long long* cand,
                 size_t cand_frame_stride, int* cand_count, uint32_t* ws_xy, uint16_t* ws_node,
                 OrbLevelKp* lkp, size_t kp_frame_stride, int* kp_count, int node_cap, int key_cap, int nframes,
                 int level_override) {
    dim3 grid(nframes, level_override >= 0 ? 1 : nlevels);
    // Workgroups of 512 threads finish a (frame, level) soonest; with a hundred frames or more per call 256-thread workgroups take 10 %
    // longer alone (74 vs 67 us at 128 frames) but leave the side stream's blur more of the chip, and the STEP is 2 % shorter
    // (0.551 vs 0.562 ms); at 64 frames they cost 5 %.
    static const int nt_env = getenv("SLAMIT_OCT_THREADS") ? atoi(getenv("SLAMIT_OCT_THREADS")) : 0;
    // ... and with a handful of frames (<= 16: at most 128 workgroups, half a chip) the pass is the level-0 workgroup's critical path:
    // 1,024 threads shorten its key sweeps (8 frames of 720p: 0.095 -> 0.069 ms; 64 frames: 512 threads are better, 0.038 vs 0.054 at VGA)
    const int nt = nt_env == 256 || nt_env == 512 || nt_env == 1024 ? nt_env : (nframes >= 96 ? 256 : nframes <= 16 ? 1024 : OCT_THREADS);
    // A big batch is bound by how many of its (frame, level) workgroups a CU holds at once, and that by their LDS: with the
    // handle's full key arrays (48 KB) three fit, with room for 2,048 keys five do, and the lists above that go through the HBM
    // workspace (L2 resident) at little cost: 0.116 -> 0.092 ms per 256 VGA frames.  (1,536 keys = six per CU: the pass alone
    // 0.088 ms, but the step 2 % LONGER -- the side stream's blur finds less of the chip.)  A few frames keep the big arrays.
    static const bool keys_env = getenv("SLAMIT_OCT_KEYS") || getenv("SLAMIT_OCT_LDS_KB");
    if (nframes >= 96 && !keys_env) key_cap = std::min(key_cap, 2048);
    if (nt == 1024)
        hipLaunchKernelGGL(octree_kernel<1024>, grid, dim3(1024), orbk_octree_smem(node_cap, key_cap), st, levels, nlevels, cand,
                           cand_frame_stride, cand_count, ws_xy, ws_node, lkp, kp_frame_stride, kp_count, node_cap,
                           key_cap, level_override);
    else if (nt == 256)
        hipLaunchKernelGGL(octree_kernel<256>, grid, dim3(256), orbk_octree_smem(node_cap, key_cap), st, levels, nlevels, cand,
                           cand_frame_stride, cand_count, ws_xy, ws_node, lkp, kp_frame_stride, kp_count, node_cap,
                           key_cap, level_override);
    else
        hipLaunchKernelGGL(octree_kernel<OCT_THREADS>, grid, dim3(OCT_THREADS), orbk_octree_smem(node_cap, key_cap), st, levels, nlevels, cand,
                           cand_frame_stride, cand_count, ws_xy, ws_node, lkp, kp_frame_stride, kp_count, node_cap,
                           key_cap, level_override);
}

void orbk_ic_angle(hipStream_t st, const OrbLevel* levels, int nlevels, const uint8_t* img0,
                   size_t img0_stride, size_t img0_frame, const uint8_t* pyr, OrbLevelKp* lkp,
                   size_t kp_frame_stride, const int* kp_count, int max_kp, int nframes) {
    const int kblocks = (max_kp + 4 * IC_KP_PER_WAVE - 1) / (4 * IC_KP_PER_WAVE);
    static const int xcd_env = getenv("SLAMIT_XCD_FRAMES") ? atoi(getenv("SLAMIT_XCD_FRAMES")) : 1;
    if (xcd_env && nframes >= 16)
        hipLaunchKernelGGL(ic_angle_kernel, dim3((unsigned)(((nframes + 7) / 8) * 8 * nlevels * kblocks)), dim3(256), 0, st, levels,
                           nlevels, img0, img0_stride, img0_frame, pyr, lkp, kp_frame_stride, kp_count, nframes, kblocks);
    else
        hipLaunchKernelGGL(ic_angle_kernel, dim3(kblocks, nlevels, nframes), dim3(256), 0, st, levels,
                           nlevels, img0, img0_stride, img0_frame, pyr, lkp, kp_frame_stride, kp_count, 0, kblocks);
}

// Strip table of blur_all_kernel: (level, bx, by0, 0) for every 64 x 64 strip of every level, level-major
int orbk_blur_tiles(const OrbLevel* host_levels, int nlevels, std::vector<uint32_t>& out) {
    out.clear();
    for (int l = 0; l < nlevels; ++l)
        for (int by = 0; by < host_levels[l].h; by += 16 * BLUR_STEPS)
            for (int bx = 0; bx < host_levels[l].w; bx += 64) { const uint32_t t[4] = {(uint32_t)l, (uint32_t)bx, (uint32_t)by, 0u}; out.insert(out.end(), t, t + 4); }
    return (int)(out.size() / 4);
}

// The same strips as two tables (both level-major): those whose columns lie inside the level (blur_stream_kernel) and the
// rest (blur_all_kernel); per level the number of entries of each.  The caller uses them when the planes are 4-byte aligned.
void orbk_blur_tiles_split(const OrbLevel* host_levels, int nlevels, std::vector<uint32_t>& stream, std::vector<uint32_t>& edge,
                           std::vector<int>& n_stream, std::vector<int>& n_edge) {
    stream.clear(); edge.clear(); n_stream.assign(nlevels, 0); n_edge.assign(nlevels, 0);
    for (int l = 0; l < nlevels; ++l)
        for (int by = 0; by < host_levels[l].h; by += 16 * BLUR_STEPS)
            for (int bx = 0; bx < host_levels[l].w; bx += 64) {
                const uint32_t t[4] = {(uint32_t)l, (uint32_t)bx, (uint32_t)by, 0u};
                const bool inside = bx >= 4 && bx + 68 <= host_levels[l].w && (host_levels[l].stride & 3) == 0;
                (inside ? stream : edge).insert((inside ? stream : edge).end(), t, t + 4);
                ++(inside ? n_stream : n_edge)[l];
            }
}

void orbk_blur_stream(hipStream_t st, const OrbLevel* levels, const uint32_t* d_tiles, int ntiles, const uint8_t* img0,
                      size_t img0_stride, size_t img0_frame, const uint8_t* pyr, uint8_t* blur, int nframes, bool edge) {
    if (ntiles <= 0) return;
    if (edge)
        hipLaunchKernelGGL(blur_stream_kernel<true>, dim3((ntiles + 15) / 16, nframes), dim3(256), 0, st, levels, reinterpret_cast<const uint4*>(d_tiles), img0,
                           img0_stride, img0_frame, pyr, blur, (unsigned)ntiles);
    else
        hipLaunchKernelGGL(blur_stream_kernel<false>, dim3((ntiles + 15) / 16, nframes), dim3(256), 0, st, levels, reinterpret_cast<const uint4*>(d_tiles), img0,
                           img0_stride, img0_frame, pyr, blur, (unsigned)ntiles);
}

void orbk_blur(hipStream_t st, const OrbLevel* levels, const uint32_t* d_tiles, int total_tiles, const uint8_t* img0,
               size_t img0_stride, size_t img0_frame, const uint8_t* pyr, uint8_t* blur, int nframes) {
    if (total_tiles <= 0) return;
    const unsigned total = (unsigned)total_tiles * (unsigned)nframes, grid = ((total + 7u) >> 3) << 3;
    const unsigned inv = (unsigned)((0x100000000ull + (unsigned)total_tiles - 1) / (unsigned)total_tiles);   // exact for logical < 2^32 / tiles
    hipLaunchKernelGGL(blur_all_kernel, dim3(grid), dim3(256), 0, st, levels, reinterpret_cast<const uint4*>(d_tiles), img0,
                       img0_stride, img0_frame, pyr, blur, (unsigned)total_tiles, inv, total);
}

void orbk_describe(hipStream_t st, const OrbLevel* levels, int nlevels, const uint8_t* blur,
                   const OrbLevelKp* lkp, size_t kp_frame_stride, const int* kp_count, slamit_kp* out_kps,
                   uint8_t* out_desc, int out_cap, int* out_n, int max_kp, int nframes) {
    const int kblocks = (max_kp + 4 * DESC_KP_PER_WAVE - 1) / (4 * DESC_KP_PER_WAVE);
    static const int xcd_env = getenv("SLAMIT_XCD_FRAMES") ? atoi(getenv("SLAMIT_XCD_FRAMES")) : 1;   // 0: the (blocks, levels, frames) grid (A/B runs)
    if (xcd_env && nframes >= 16)
        hipLaunchKernelGGL(describe_kernel, dim3((unsigned)(((nframes + 7) / 8) * 8 * nlevels * kblocks)), dim3(256), 0, st, levels,
                           nlevels, blur, lkp, kp_frame_stride, kp_count, out_kps, out_desc, out_cap, out_n, nframes, kblocks);
    else
        hipLaunchKernelGGL(describe_kernel, dim3(kblocks, nlevels, nframes), dim3(256), 0, st, levels,
                           nlevels, blur, lkp, kp_frame_stride, kp_count, out_kps, out_desc, out_cap, out_n, 0, kblocks);
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
