// orb_api.hip — C-ABI of the ORB extractor (include/slamit.h, slamit_orb_*): handle, HBM layout,
// coefficient tables and the launch sequence.  Mirrors the interface of ORB_SLAM2::ORBextractor
// (include/ORBextractor.h:45-111, src/ORBextractor.cc:415-482,1064-1168).  No CPU compute path:
// every entry point fails with SLAMIT_ERR_DEVICE when no HIP device is usable.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/slamit.h"
#include "orb_types.h"
#include "slamit_internal.h"

// kernels (orb_kernels.hip)
hipError_t orbk_upload_pattern(hipStream_t st);
void orbk_resize(hipStream_t st, const uint8_t* src, int sw, int sh, size_t sstride, size_t sframe,
                 uint8_t* dst, int dw, int dh, size_t dstride, size_t dframe, const int* xofs,
                 const short* ialpha, const int* yofs, const short* ibeta, int nframes);
bool orbk_resize_tables(int dw, int dh, int sw, int sh, const int* xofs, const short* ialpha, const int* yofs,
                        const short* ibeta, std::vector<uint32_t>& col, std::vector<uint32_t>& row);
void orbk_resize_rows4(hipStream_t st, const uint8_t* src, size_t sstride, size_t sframe, int sh, uint8_t* dst, int dw, int dh,
                       size_t dstride, size_t dframe, const uint32_t* d_col, const uint32_t* d_row, int nframes);
bool orbk_resize_tables8(int dw, int sw, size_t dstride, const int* xofs, const short* ialpha, std::vector<uint32_t>& col);
void orbk_resize_rows8(hipStream_t st, const uint8_t* src, size_t sstride, size_t sframe, int sh, uint8_t* dst, int dw, int dh,
                       size_t dstride, size_t dframe, const uint32_t* d_col8, const uint32_t* d_row, int nframes);
hipError_t orbk_pyramid_prepare(int smem_bytes);
void orbk_pyramid(hipStream_t st, const OrbLevel* levels, int nlevels, const PyrBox* boxes, const PyrTabs* tabs,
                  int nregions, const uint8_t* img0, size_t img0_stride, size_t img0_frame, uint8_t* pyr, int bufA_bytes,
                  int smem_bytes, int nframes, int l_first, int l_last, int threads);
size_t orbk_fast_smem(int max_wcell, int max_hcell);
hipError_t orbk_fast_prepare(int max_wcell, int max_hcell);
int orbk_fast_cells(const OrbLevel* host_levels, int nlevels, std::vector<uint32_t>& out);
void orbk_fast(hipStream_t st, const OrbLevel* host_levels, int nlevels, const uint32_t* d_cells, int cells_per_frame, const uint8_t* img0,
               size_t img0_stride, size_t img0_frame, const uint8_t* pyr, unsigned long long* cand,
               size_t cand_frame_stride, int* cand_count, int iniTh, int minTh, int max_wcell, int max_hcell, int nframes);
int orbk_octree_key_cap(int node_cap, int width, int height);
size_t orbk_octree_smem(int node_cap, int key_cap);
hipError_t orbk_octree_prepare(int node_cap, int key_cap);
void orbk_octree(hipStream_t st, const OrbLevel* levels, int nlevels, const unsigned long long* cand,
                 size_t cand_frame_stride, int* cand_count, uint32_t* ws_xy, uint16_t* ws_node,
                 OrbLevelKp* lkp, size_t kp_frame_stride, int* kp_count, int node_cap, int key_cap, int nframes,
                 int level_override);
void orbk_ic_angle(hipStream_t st, const OrbLevel* levels, int nlevels, const uint8_t* img0, size_t img0_stride,
                   size_t img0_frame, const uint8_t* pyr, OrbLevelKp* lkp, size_t kp_frame_stride,
                   const int* kp_count, int max_kp, int nframes);
int orbk_blur_tiles(const OrbLevel* host_levels, int nlevels, std::vector<uint32_t>& out);
void orbk_blur_tiles_split(const OrbLevel* host_levels, int nlevels, std::vector<uint32_t>& stream, std::vector<uint32_t>& edge,
                           std::vector<int>& n_stream, std::vector<int>& n_edge);
void orbk_blur_stream(hipStream_t st, const OrbLevel* levels, const uint32_t* d_tiles, int ntiles, const uint8_t* img0,
                      size_t img0_stride, size_t img0_frame, const uint8_t* pyr, uint8_t* blur, int nframes, bool edge);
void orbk_blur(hipStream_t st, const OrbLevel* levels, const uint32_t* d_tiles, int total_tiles, const uint8_t* img0,
               size_t img0_stride, size_t img0_frame, const uint8_t* pyr, uint8_t* blur, int nframes);
void orbk_describe(hipStream_t st, const OrbLevel* levels, int nlevels, const uint8_t* blur,
                   const OrbLevelKp* lkp, size_t kp_frame_stride, const int* kp_count, slamit_kp* out_kps,
                   uint8_t* out_desc, int out_cap, int* out_n, int max_kp, int nframes);
void orbk_pad(hipStream_t st, const uint8_t* src, int w, int h, size_t sstride, uint8_t* dst);
void orbk_decode_candidates(hipStream_t st, const OrbLevel* levels, int level, const unsigned long long* K, int n,
                            unsigned long long* order_out, int* xys);

namespace {

inline int cv_round(double v) { return (int)lrint(v); }  // cvRound: half-to-even
inline size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

inline short sat_short(float v) {
    int iv = cv_round((double)v);
    return (short)(iv < -32768 ? -32768 : iv > 32767 ? 32767 : iv);
}

// per-axis tables of cv::resize INTER_LINEAR 8U (fixed point, 11 bits); `clampx` applies the
// x-axis rule (offset clamped and weight zeroed at both ends), the y axis keeps its weights
void resize_axis(int dn, int sn, bool clampx, std::vector<int>& ofs, std::vector<short>& coef) {
    double inv_scale = (double)dn / sn;
    double scale = 1. / inv_scale;
    ofs.resize(dn);
    coef.resize(2 * (size_t)dn);
    for (int d = 0; d < dn; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floor(f);
        f -= s;
        if (clampx) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= sn - 1) { f = 0; s = sn - 1; }
        }
        ofs[d] = s;
        coef[2 * d] = sat_short((1.f - f) * 2048.f);
        coef[2 * d + 1] = sat_short(f * 2048.f);
    }
}

}  // namespace

struct slamit_orb {
    slamit_orb_params p;
    int device;
    hipStream_t stream;
    hipStream_t stream_b;     // side stream: the blur of a call runs beside its FAST / octree / orientation launches
    hipEvent_t ev_pyr, ev_mid, ev_blur;
    int blur_split;       // levels 0 .. blur_split are blurred beside the tail of the pyramid chain (-1: all after FAST)
    int overlap;              // 0: everything on one stream (SLAMIT_ORB_SERIAL=1)
    int nlevels;
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> per_level;
    std::vector<OrbLevel> levels;
    int cells_per_frame, blur_tiles, node_cap, oct_key_cap, max_kp_level, max_out, max_wcell, max_hcell;
    size_t pyr_frame_total, blur_frame_total;
    size_t cand_frame_stride, kp_frame_stride;
    // device memory
    OrbLevel* d_levels;
    uint8_t* d_pyr;
    uint8_t* d_blur;
    unsigned long long* d_cand;
    uint32_t* d_ws_xy;
    uint16_t* d_ws_node;
    uint32_t* d_blur_tiles;   // blur strip table (orbk_blur_tiles), blur_tiles entries of 4 words
    uint32_t* d_blur_str; uint32_t* d_blur_edge;   // the same strips split: columns inside the level (blur_stream_kernel) / the rest (orbk_blur_tiles_split)
    int blur_str_base[ORB_MAX_LEVELS + 1], blur_edge_base[ORB_MAX_LEVELS + 1];   // first entry of a level in each table (last: the totals)
    bool blur_stream_on;
    uint32_t* d_cells;   // FAST cell table (orbk_fast_cells), fast_cells entries of 4 words
    int fast_cells;
    int* d_counts;  // cand_count [max_batch][nlevels][ORB_CC_PAD], then kp_count [max_batch][nlevels]
    bool counters_clean;   // every cand_count is zero (left so by the last call's octree pass)
    OrbLevelKp* d_lkp;
    int* d_tab_i[ORB_MAX_LEVELS][2];      // xofs, yofs per level (level >= 1)
    short* d_tab_s[ORB_MAX_LEVELS][2];    // ialpha, ibeta
    uint32_t* d_rs_col[ORB_MAX_LEVELS];   // resize_rows4_kernel tables (orbk_resize_tables)
    uint32_t* d_rs_row[ORB_MAX_LEVELS];
    uint32_t* d_rs_col8[ORB_MAX_LEVELS];  // resize_rows8_kernel column tables; null where the level's geometry does not fit it
    int pyr_mode;                         // 0 per-level rows4 (default), 1 fused segments (SLAMIT_PYR_FUSED)
    PyrBox* d_boxes;                      // fused pyramid: [nregions][nlevels]
    PyrTabs* d_tabs;                      // [nlevels]
    struct PyrSeg { int first, last, nregions, box_off, bufA, smem, threads; };
    std::vector<PyrSeg> pyr_segs;          // empty = per-level fallback kernel
    // staging for the host-pointer entry points
    uint8_t* d_in;
    size_t d_in_stride, d_in_frame;
    slamit_kp* d_out_kps;
    uint8_t* d_out_desc;
    int* d_out_n;
    uint8_t* h_out;   // pinned: [n per frame | keypoints | descriptors] of a whole batch, so the readback is one stream op chain + one sync
    uint8_t* d_scratch;  // padded plane / debug scratch
    size_t scratch_bytes;
    // optional per-stage hipEvent timing (slamit_orb_profile)
    int prof_on;
    long prof_call;                       // extract calls since profiling was switched on
    std::vector<hipEvent_t> prof_ev;   // pairs (begin, end)
    std::vector<int> prof_stage;       // stage id of each pair
    // last call (for slamit_orb_level / debug getters)
    const uint8_t* last_img0;
    size_t last_stride, last_frame;
    int last_nframes;
};

static void orb_free(slamit_orb* h) {
    if (!h) return;
    SlamitDeviceGuard guard(h->device);
    hipFree(h->d_levels); hipFree(h->d_pyr); hipFree(h->d_blur); hipFree(h->d_cand); hipFree(h->d_ws_xy);
    hipFree(h->d_ws_node); hipFree(h->d_counts); hipFree(h->d_cells); hipFree(h->d_blur_tiles); hipFree(h->d_blur_str); hipFree(h->d_blur_edge); hipFree(h->d_lkp); hipFree(h->d_in); hipFree(h->d_out_kps);
    hipFree(h->d_out_desc); hipFree(h->d_out_n); if (h->h_out) hipHostFree(h->h_out); hipFree(h->d_scratch); hipFree(h->d_boxes); hipFree(h->d_tabs);
    for (int l = 0; l < ORB_MAX_LEVELS; ++l)
        for (int a = 0; a < 2; ++a) { hipFree(h->d_tab_i[l][a]); hipFree(h->d_tab_s[l][a]); }
    for (int l = 0; l < ORB_MAX_LEVELS; ++l) { hipFree(h->d_rs_col[l]); hipFree(h->d_rs_row[l]); hipFree(h->d_rs_col8[l]); }
    for (hipEvent_t e : h->prof_ev) hipEventDestroy(e);
    if (h->ev_pyr) hipEventDestroy(h->ev_pyr);
    if (h->ev_mid) hipEventDestroy(h->ev_mid);
    if (h->ev_blur) hipEventDestroy(h->ev_blur);
    if (h->stream_b) hipStreamDestroy(h->stream_b);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

extern "C" {

int slamit_orb_create(const slamit_orb_params* p, int device, slamit_orb** out) {
    if (!p || !out) return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_create: null argument");
    *out = nullptr;
    if (p->nlevels < 1 || p->nlevels > ORB_MAX_LEVELS || p->nfeatures < 1 || !(p->scale_factor > 1.f) ||
        p->ini_th_fast < 1 || p->ini_th_fast > 255 || p->min_th_fast < 1 || p->min_th_fast > 255 ||
        p->width < 0 || p->height < 0 || p->max_batch < 1)
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_create: parameter out of range");
    SLAMIT_USE_DEVICE(device);
    slamit_orb* h = new slamit_orb();
    h->p = *p;
    h->device = device;
    h->stream = nullptr;
    h->last_img0 = nullptr; h->last_nframes = 0;
    const int nl = h->nlevels = p->nlevels;

    // ---- scale tables and quotas (ORBextractor.cc:422-455; scaleFactor is a double member) ----
    const double scaleFactor = (double)p->scale_factor;
    h->scale.assign(nl, 1.f); h->sigma2.assign(nl, 1.f); h->inv_scale.assign(nl, 1.f); h->inv_sigma2.assign(nl, 1.f);
    for (int i = 1; i < nl; ++i) {
        h->scale[i] = (float)(h->scale[i - 1] * scaleFactor);
        h->sigma2[i] = h->scale[i] * h->scale[i];
    }
    for (int i = 0; i < nl; ++i) { h->inv_scale[i] = 1.0f / h->scale[i]; h->inv_sigma2[i] = 1.0f / h->sigma2[i]; }
    h->per_level.assign(nl, 0);
    {
        float factor = (float)(1.0f / scaleFactor);
        float nDesired = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nl));
        int sum = 0;
        for (int l = 0; l < nl - 1; ++l) {
            h->per_level[l] = cv_round(nDesired);
            sum += h->per_level[l];
            nDesired *= factor;
        }
        h->per_level[nl - 1] = std::max(p->nfeatures - sum, 0);
    }
    h->max_out = p->nfeatures + 3 * nl;

    // ---- level geometry (ORBextractor.cc:1143-1147, 789-803, 556-571) ----
    h->levels.assign(nl, OrbLevel());
    size_t pyr_off = 0, blur_off = 0, cand_off = 0;
    int kp_off = 0, cell_base = 0, blur_tiles = 0;
    h->node_cap = 8; h->max_kp_level = 1;
    int sum_cap = 0;
    const bool empty = p->width == 0 || p->height == 0;
    for (int l = 0; l < nl && !empty; ++l) {
        OrbLevel& L = h->levels[l];
        float sc = h->inv_scale[l];
        L.w = cv_round((float)p->width * sc);
        L.h = cv_round((float)p->height * sc);
        L.stride = (int)round_up((size_t)std::max(L.w, 1), 64);
        L.quota = h->per_level[l];
        L.plane_bytes = (size_t)L.stride * std::max(L.h, 1);
        L.blur_bytes = L.plane_bytes;
        if (l >= 1) { L.plane_off = pyr_off; pyr_off += round_up(L.plane_bytes, 256); } else L.plane_off = 0;
        L.blur_off = blur_off; blur_off += round_up(L.blur_bytes, 256);
        L.maxBorderX = L.w - ORB_MIN_BORDER; L.maxBorderY = L.h - ORB_MIN_BORDER;
        const float width = (float)(L.maxBorderX - ORB_MIN_BORDER), height = (float)(L.maxBorderY - ORB_MIN_BORDER);
        L.nCols = (int)(width / 30.f); L.nRows = (int)(height / 30.f);
        if (L.w < 1 || L.h < 1 || L.nCols < 1 || L.nRows < 1) {
            // the reference divides by zero on such a level; refuse the geometry
            orb_free(h);
            return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_create: pyramid level smaller than one 30x30 FAST cell");
        }
        L.wCell = (int)ceil(width / L.nCols); L.hCell = (int)ceil(height / L.nRows);
        L.cell_base = cell_base; L.ncells = L.nCols * L.nRows; cell_base += L.ncells;
        h->max_wcell = std::max(h->max_wcell, L.wCell); h->max_hcell = std::max(h->max_hcell, L.hCell);
        L.blur_tile_base = blur_tiles; blur_tiles += ((L.w + 63) / 64) * ((L.h + 63) / 64);  // 64x64 strips, four 16-row steps each
        L.cand_cap = L.ncells * ((L.wCell + 1) / 2) * ((L.hCell + 1) / 2);  // NMS: <= 1 per 2x2 in a cell
        L.cand_off = cand_off; cand_off += round_up((size_t)L.cand_cap, 64);
        const int bw = L.maxBorderX - ORB_MIN_BORDER, bh = L.maxBorderY - ORB_MIN_BORDER;
        L.nIni = (int)round(static_cast<float>(bw) / bh);
        if (L.nIni < 1 || L.nIni > ORB_MAX_ROOTS) {
            orb_free(h);
            return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_create: unsupported aspect ratio (octree roots)");
        }
        L.hX = static_cast<float>(bw) / L.nIni;
        for (int i = 0; i < L.nIni; ++i) {
            L.rootUL[i] = (int)(L.hX * static_cast<float>(i));
            L.rootUR[i] = (int)(L.hX * static_cast<float>(i + 1));
        }
        L.boxH = bh;
        L.kp_cap = std::max(L.quota, 4 * L.nIni) + 4;
        L.kp_off = kp_off; kp_off += L.kp_cap;
        L.scale = h->scale[l];
        L.patch_size = (float)(int)(31 * h->scale[l]);
        h->node_cap = std::max(h->node_cap, L.kp_cap);
        h->max_kp_level = std::max(h->max_kp_level, L.kp_cap);
        sum_cap += L.kp_cap;
    }
    h->max_out = std::max(h->max_out, sum_cap);
    h->cells_per_frame = cell_base;
    h->blur_tiles = blur_tiles;
    h->pyr_frame_total = pyr_off; h->blur_frame_total = blur_off;
    h->cand_frame_stride = cand_off; h->kp_frame_stride = (size_t)kp_off;
    // frames are the outer dimension of every per-frame array: plane(level, f) = base + level_off + f*frame_total
    for (int l = 0; l < nl; ++l) {
        h->levels[l].plane_bytes = h->pyr_frame_total;   // stride between frames of the same level
        h->levels[l].blur_bytes = h->blur_frame_total;
    }
    h->oct_key_cap = orbk_octree_key_cap(h->node_cap, p->width, p->height);
    if (orbk_octree_smem(h->node_cap, h->oct_key_cap) > 160 * 1024 - 1024 || h->node_cap >= 4096) {   // labels carry the node in 12 bits
        orb_free(h);
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_create: nfeatures too large for the LDS octree");
    }

    const size_t B = (size_t)p->max_batch;
    hipError_t e = hipSuccess;
#define ALLOC(ptr, bytes) if (e == hipSuccess) e = hipMalloc((void**)&(ptr), std::max<size_t>((bytes), 256))
    ALLOC(h->d_levels, sizeof(OrbLevel) * nl);
    ALLOC(h->d_pyr, h->pyr_frame_total * B);
    ALLOC(h->d_blur, h->blur_frame_total * B + 256);   // + slack: the descriptor kernel stages whole dwords up to 6 bytes past a row end
    ALLOC(h->d_cand, sizeof(unsigned long long) * h->cand_frame_stride * B);
    ALLOC(h->d_ws_xy, sizeof(uint32_t) * h->cand_frame_stride * B);
    ALLOC(h->d_ws_node, sizeof(uint16_t) * h->cand_frame_stride * B);
    ALLOC(h->d_counts, sizeof(int) * ((ORB_CC_PAD + 1) * B * nl + 2 * ORB_CC_PAD));
    ALLOC(h->d_lkp, sizeof(OrbLevelKp) * h->kp_frame_stride * B);
    h->d_in_stride = round_up((size_t)std::max(p->width, 1), 64);
    h->d_in_frame = h->d_in_stride * std::max(p->height, 1);
    ALLOC(h->d_in, h->d_in_frame * B);
    ALLOC(h->d_out_kps, sizeof(slamit_kp) * (size_t)h->max_out * B);
    ALLOC(h->d_out_desc, (size_t)SLAMIT_DESC_BYTES * h->max_out * B);
    ALLOC(h->d_out_n, sizeof(int) * B);
    if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_out, 256 + (size_t)B * 256 + (sizeof(slamit_kp) + SLAMIT_DESC_BYTES) * (size_t)h->max_out * B, hipHostMallocDefault);
    h->scratch_bytes = std::max<size_t>((size_t)(p->width + 38) * (p->height + 38),
                                        (sizeof(unsigned long long) + 3 * sizeof(int)) * (h->cand_frame_stride + 64));
    ALLOC(h->d_scratch, h->scratch_bytes);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream_b, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_pyr, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_mid, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_blur, hipEventDisableTiming);
    h->overlap = getenv("SLAMIT_ORB_SERIAL") ? 0 : 1;
    if (e == hipSuccess && !empty) e = hipMemcpy(h->d_levels, h->levels.data(), sizeof(OrbLevel) * nl, hipMemcpyHostToDevice);
    {
        std::vector<uint32_t> cells;
        h->fast_cells = empty ? 0 : orbk_fast_cells(h->levels.data(), nl, cells);
        ALLOC(h->d_cells, sizeof(uint32_t) * std::max<size_t>(cells.size(), 8));
        if (e == hipSuccess && !cells.empty()) e = hipMemcpy(h->d_cells, cells.data(), sizeof(uint32_t) * cells.size(), hipMemcpyHostToDevice);
        std::vector<uint32_t> bt;
        h->blur_tiles = empty ? 0 : orbk_blur_tiles(h->levels.data(), nl, bt);
        ALLOC(h->d_blur_tiles, sizeof(uint32_t) * std::max<size_t>(bt.size(), 4));
        if (e == hipSuccess && !bt.empty()) e = hipMemcpy(h->d_blur_tiles, bt.data(), sizeof(uint32_t) * bt.size(), hipMemcpyHostToDevice);
        {
            std::vector<uint32_t> ts, te;
            std::vector<int> ns, ne;
            if (!empty) orbk_blur_tiles_split(h->levels.data(), nl, ts, te, ns, ne);
            for (int l = 0, a = 0, b = 0; l <= nl; ++l) { h->blur_str_base[l] = a; h->blur_edge_base[l] = b; if (l < nl && !empty) { a += ns[l]; b += ne[l]; } }
            ALLOC(h->d_blur_str, sizeof(uint32_t) * std::max<size_t>(ts.size(), 4)); ALLOC(h->d_blur_edge, sizeof(uint32_t) * std::max<size_t>(te.size(), 4));
            if (e == hipSuccess && !ts.empty()) e = hipMemcpy(h->d_blur_str, ts.data(), sizeof(uint32_t) * ts.size(), hipMemcpyHostToDevice);
            if (e == hipSuccess && !te.empty()) e = hipMemcpy(h->d_blur_edge, te.data(), sizeof(uint32_t) * te.size(), hipMemcpyHostToDevice);
            h->blur_stream_on = !empty && !(getenv("SLAMIT_BLUR_NO_STREAM") && atoi(getenv("SLAMIT_BLUR_NO_STREAM")));   // A/B runs: the tile kernel everywhere
        }
    }
    bool rows4_ok = true;
    for (int l = 1; l < nl && e == hipSuccess && !empty; ++l) {
        std::vector<int> xo, yo;
        std::vector<short> xa, ya;
        resize_axis(h->levels[l].w, h->levels[l - 1].w, true, xo, xa);
        resize_axis(h->levels[l].h, h->levels[l - 1].h, false, yo, ya);
        ALLOC(h->d_tab_i[l][0], xo.size() * 4); ALLOC(h->d_tab_i[l][1], yo.size() * 4);
        ALLOC(h->d_tab_s[l][0], xa.size() * 2); ALLOC(h->d_tab_s[l][1], ya.size() * 2);
        if (e == hipSuccess) e = hipMemcpy(h->d_tab_i[l][0], xo.data(), xo.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(h->d_tab_i[l][1], yo.data(), yo.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(h->d_tab_s[l][0], xa.data(), xa.size() * 2, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(h->d_tab_s[l][1], ya.data(), ya.size() * 2, hipMemcpyHostToDevice);
        std::vector<uint32_t> ct, rt;
        if (!orbk_resize_tables(h->levels[l].w, h->levels[l].h, h->levels[l - 1].w, h->levels[l - 1].h, xo.data(), xa.data(), yo.data(),
                                ya.data(), ct, rt))
            rows4_ok = false;
        ALLOC(h->d_rs_col[l], ct.size() * 4); ALLOC(h->d_rs_row[l], rt.size() * 4);
        if (e == hipSuccess) e = hipMemcpy(h->d_rs_col[l], ct.data(), ct.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(h->d_rs_row[l], rt.data(), rt.size() * 4, hipMemcpyHostToDevice);
        static const bool no_rows8 = getenv("SLAMIT_RESIZE_NO8") && atoi(getenv("SLAMIT_RESIZE_NO8"));   // A/B runs: the four-pixel kernel everywhere
        std::vector<uint32_t> c8;
        if (!no_rows8 && orbk_resize_tables8(h->levels[l].w, h->levels[l - 1].w, (size_t)h->levels[l].stride, xo.data(), xa.data(), c8)) {
            ALLOC(h->d_rs_col8[l], c8.size() * 4);
            if (e == hipSuccess) e = hipMemcpy(h->d_rs_col8[l], c8.data(), c8.size() * 4, hipMemcpyHostToDevice);
        }
    }
    // ---- fused pyramid: the levels are built in SEGMENTS (default: 0 -> 1,2 | 2 -> 3,4 | 4 -> 5..): one launch
    // per segment, each workgroup reads its patch of the segment's first level and produces the following levels
    // out of LDS.  Short segments keep the halo (pixels computed only because a deeper level needs them) small;
    // per region and level the boxes say what it stores ("own") and what it has to compute ("need").
    if (e == hipSuccess && !empty && nl > 1) {
        std::vector<std::vector<int> > XO(nl), YO(nl);
        for (int l = 1; l < nl; ++l) {
            std::vector<short> dummy;
            resize_axis(h->levels[l].w, h->levels[l - 1].w, true, XO[l], dummy);
            resize_axis(h->levels[l].h, h->levels[l - 1].h, false, YO[l], dummy);
        }
        // Default: ONE segment, regions of 128x96 level-0 pixels (what the unaligned / large-scale-factor fallback was
        // tuned for).  SLAMIT_PYR_SEGS="2,4" / SLAMIT_PYR_TILE="64x48" (tile at the segment's last level) are
        // diagnostic knobs for experiments with shorter chains.
        std::vector<int> cuts;
        if (const char* sv = getenv("SLAMIT_PYR_SEGS"))
            for (const char* q = sv; *q;) { int v = atoi(q); if (v > 0 && v < nl - 1) cuts.push_back(v); while (*q && *q != ',') ++q; if (*q) ++q; }
        std::sort(cuts.begin(), cuts.end());
        cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
        int tw = 0, th = 0;
        if (const char* sv = getenv("SLAMIT_PYR_TILE")) { if (sscanf(sv, "%dx%d", &tw, &th) != 2 || tw < 8 || th < 8) tw = th = 0; }
        std::vector<PyrBox> boxes;
        bool okb = true;
        // plans one segment first -> last with regions of tw x th pixels at its last level (0 = 128x96 level-0 pixels)
        auto plan = [&](int first, int last, int tw, int th, slamit_orb::PyrSeg& seg) -> bool {
            bool ok = true;
            seg.first = first; seg.last = last; seg.box_off = (int)boxes.size();
            const OrbLevel& LL = h->levels[last];
            const int GX = tw ? std::max(1, (LL.w + tw - 1) / tw) : std::max(1, (p->width + 127) / 128);
            const int GY = th ? std::max(1, (LL.h + th - 1) / th) : std::max(1, (p->height + 95) / 96);
            seg.nregions = GX * GY;
            boxes.resize(boxes.size() + (size_t)GX * GY * nl);
            size_t capA = 16, capB = 16;
            for (int gy = 0; gy < GY && ok; ++gy)
                for (int gx = 0; gx < GX && ok; ++gx) {
                    PyrBox* B = &boxes[seg.box_off + ((size_t)gy * GX + gx) * nl];
                    for (int l = first; l <= last; ++l) {
                        const OrbLevel& L = h->levels[l];
                        B[l].ox0 = (int16_t)((long)gx * L.w / GX); B[l].ox1 = (int16_t)((long)(gx + 1) * L.w / GX);
                        B[l].oy0 = (int16_t)((long)gy * L.h / GY); B[l].oy1 = (int16_t)((long)(gy + 1) * L.h / GY);
                        if (l == first) { B[l].ox0 = B[l].ox1 = B[l].oy0 = B[l].oy1 = 0; }  // the segment's source is only read
                        else if (B[l].ox1 <= B[l].ox0 || B[l].oy1 <= B[l].oy0) ok = false;
                    }
                    B[last].nx0 = B[last].ox0; B[last].nx1 = B[last].ox1;
                    B[last].ny0 = B[last].oy0; B[last].ny1 = B[last].oy1;
                    for (int l = last; l > first && ok; --l) {
                        const int sw = h->levels[l - 1].w, sh = h->levels[l - 1].h;
                        int sx0 = XO[l][B[l].nx0], sx1 = std::min(XO[l][B[l].nx1 - 1] + 1, sw - 1) + 1;
                        int sy0 = std::min(std::max(YO[l][B[l].ny0], 0), sh - 1);
                        int sy1 = std::min(std::max(YO[l][B[l].ny1 - 1] + 1, 0), sh - 1) + 1;
                        if (l - 1 > first) {
                            sx0 = std::min(sx0, (int)B[l - 1].ox0); sx1 = std::max(sx1, (int)B[l - 1].ox1);
                            sy0 = std::min(sy0, (int)B[l - 1].oy0); sy1 = std::max(sy1, (int)B[l - 1].oy1);
                        }
                        if (l - 1 == first) sx0 &= ~3;  // dword-aligned source patch
                        B[l - 1].nx0 = (int16_t)sx0; B[l - 1].nx1 = (int16_t)sx1; B[l - 1].ny0 = (int16_t)sy0; B[l - 1].ny1 = (int16_t)sy1;
                    }
                    for (int l = first; l <= last; ++l) {
                        if (l > first && (B[l].nx1 - B[l].nx0 > 256 || B[l].ny1 - B[l].ny0 > 256)) ok = false;  // <= 4 columns per lane, row tables of 256
                        size_t bytes = (size_t)(((B[l].nx1 - B[l].nx0) + 3) & ~3) * (B[l].ny1 - B[l].ny0);
                        if ((l - first) & 1) capB = std::max(capB, bytes); else capA = std::max(capA, bytes);
                    }
                }
            seg.bufA = (int)round_up(capA, 16);
            seg.smem = seg.bufA + (int)round_up(capB, 16);
            if (seg.smem > 150 * 1024) ok = false;
            seg.threads = tw && (size_t)tw * th <= 64 * 48 ? 256 : 512;
            return ok;
        };
        int first = 0;
        h->pyr_segs.clear();
        for (size_t si = 0; si <= cuts.size() && okb; ++si) {
            const int last = si < cuts.size() ? cuts[si] : nl - 1;
            slamit_orb::PyrSeg seg;
            okb = plan(first, last, tw, th, seg);
            h->pyr_segs.push_back(seg);
            first = last;
        }
        h->pyr_mode = (getenv("SLAMIT_PYR_FUSED") || !rows4_ok) ? 1 : 0;
        h->blur_split = getenv("SLAMIT_BLUR_SPLIT") ? atoi(getenv("SLAMIT_BLUR_SPLIT")) : 1;   // (2 until the pyramid chain got its wide loads: its tail is shorter now)
        if (getenv("SLAMIT_PYR_PER_LEVEL")) { okb = false; h->pyr_mode = 1; }   // diagnostic: force the old per-level kernel
        if (!okb) h->pyr_segs.clear();                      // fall back to the per-level kernel
        if (!h->pyr_segs.empty()) {
            std::vector<PyrTabs> tabs(nl);
            for (int l = 0; l < nl; ++l) { tabs[l].xofs = h->d_tab_i[l][0]; tabs[l].ialpha = h->d_tab_s[l][0]; tabs[l].yofs = h->d_tab_i[l][1]; tabs[l].ibeta = h->d_tab_s[l][1]; }
            ALLOC(h->d_boxes, sizeof(PyrBox) * boxes.size());
            ALLOC(h->d_tabs, sizeof(PyrTabs) * nl);
            if (e == hipSuccess) e = hipMemcpy(h->d_boxes, boxes.data(), sizeof(PyrBox) * boxes.size(), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(h->d_tabs, tabs.data(), sizeof(PyrTabs) * nl, hipMemcpyHostToDevice);
            int smax = 0;
            for (const slamit_orb::PyrSeg& sg : h->pyr_segs) smax = std::max(smax, sg.smem);
            if (e == hipSuccess) e = orbk_pyramid_prepare(smax);
        }
    }
#undef ALLOC
    if (e == hipSuccess) e = orbk_upload_pattern(h->stream);
    if (e == hipSuccess) e = orbk_octree_prepare(h->node_cap, h->oct_key_cap);
    if (e == hipSuccess && !empty) e = orbk_fast_prepare(h->max_wcell, h->max_hcell);
    if (e == hipSuccess && h->d_counts) e = hipMemset(h->d_counts, 0, sizeof(int) * ((ORB_CC_PAD + 1) * B * nl + 2 * ORB_CC_PAD));
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        orb_free(h);
        return slamit_fail_hip(e, "slamit_orb_create");
    }
    *out = h;
    return SLAMIT_OK;
}

void slamit_orb_destroy(slamit_orb* h) { orb_free(h); }

int slamit_orb_tables(const slamit_orb* h, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                      int32_t* features_per_level) {
    if (!h) return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_tables: null handle");
    for (int i = 0; i < h->nlevels; ++i) {
        if (scale) scale[i] = h->scale[i];
        if (inv_scale) inv_scale[i] = h->inv_scale[i];
        if (sigma2) sigma2[i] = h->sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = h->inv_sigma2[i];
        if (features_per_level) features_per_level[i] = h->per_level[i];
    }
    return SLAMIT_OK;
}

int slamit_orb_max_keypoints(const slamit_orb* h) { return h ? h->max_out : 0; }

// stage ids reported by slamit_orb_profile
enum { ST_RESIZE = 0, ST_FAST, ST_OCTREE, ST_ANGLE, ST_BLUR, ST_DESCRIBE, ST_COUNT };

static void prof_mark(slamit_orb* h, hipStream_t st, int stage, bool begin) {
    if (!h->prof_on || h->prof_ev.size() >= 2 * 16384) return;
    if (h->prof_on >= 2 && (stage != ST_FAST || h->prof_call % (h->prof_on - 1) != 0)) return;   // dominant kernel, sampled
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return;
    hipEventRecord(e, st);
    h->prof_ev.push_back(e);
    if (begin) h->prof_stage.push_back(stage);
}

int slamit_orb_extract_batch_dev(slamit_orb* h, const uint8_t* d_gray, size_t stride, size_t frame_stride,
                                 int nframes, slamit_kp* d_kps, uint8_t* d_desc, int cap, int32_t* d_n_out,
                                 void* stream) {
    if (!h || !d_n_out || nframes < 0) return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_extract_batch_dev: bad argument");
    if (nframes > h->p.max_batch) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_orb_extract_batch_dev: nframes > max_batch");
    if (nframes == 0) return SLAMIT_OK;
    SLAMIT_USE_DEVICE(h->device);
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    const int nl = h->nlevels;
    if (h->p.width == 0 || h->p.height == 0) {  // ORBextractor.cc:1068: empty image -> nothing
        HIP_TRY(hipMemsetAsync(d_n_out, 0, sizeof(int) * nframes, st));
        h->last_nframes = 0;
        return SLAMIT_OK;
    }
    if (!d_gray || !d_kps || !d_desc) return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_extract_batch_dev: null buffer");
    if (cap < h->max_out) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_orb_extract_batch_dev: cap < slamit_orb_max_keypoints()");
    if (stride < (size_t)h->p.width || (nframes > 1 && frame_stride < stride * (size_t)h->p.height))
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_extract_batch_dev: stride smaller than the frame");
    if (stride >= ((size_t)1 << 24)) return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_extract_batch_dev: row pitch of 16 MiB or more");   // kernels address rows with 24-bit multiplies
    int* cand_count = h->d_counts;
    int* kp_count = h->d_counts + (size_t)h->p.max_batch * nl * ORB_CC_PAD;
    // the candidate counters are zero between calls: the octree pass consumes and re-zeroes them.  Only a call that
    // follows a failed one (or the first) clears them itself.
    if (!h->counters_clean) HIP_TRY(hipMemsetAsync(h->d_counts, 0, sizeof(int) * (size_t)h->p.max_batch * nl * ORB_CC_PAD, st));
    h->counters_clean = false;
    // the blur of levels [l0, l1): every strip streams down its columns (blur_stream_kernel; the strips at the left / right edge
    // in their own launch); when the caller's level-0 plane is not 4-byte aligned everything takes the tile kernel
    const bool blur_src_aligned = ((((uintptr_t)d_gray) | stride | frame_stride) & 3) == 0;
    auto launch_blur = [&](hipStream_t bs, int l0, int l1) {
        if (h->blur_stream_on && blur_src_aligned) {
            orbk_blur_stream(bs, h->d_levels, h->d_blur_str + 4 * (size_t)h->blur_str_base[l0], h->blur_str_base[l1] - h->blur_str_base[l0], d_gray, stride,
                             frame_stride, h->d_pyr, h->d_blur, nframes, false);
            orbk_blur_stream(bs, h->d_levels, h->d_blur_edge + 4 * (size_t)h->blur_edge_base[l0], h->blur_edge_base[l1] - h->blur_edge_base[l0], d_gray, stride,
                             frame_stride, h->d_pyr, h->d_blur, nframes, true);
        } else {
            const int t0 = h->levels[l0].blur_tile_base, t1 = l1 < nl ? h->levels[l1].blur_tile_base : h->blur_tiles;
            orbk_blur(bs, h->d_levels, h->d_blur_tiles + 4 * (size_t)t0, t1 - t0, d_gray, stride, frame_stride, h->d_pyr, h->d_blur, nframes);
        }
    };
    // K1: pyramid, level l from level l-1
    prof_mark(h, st, ST_RESIZE, true);
    const bool src0_aligned = ((((uintptr_t)d_gray) | stride | frame_stride) & 3) == 0;
    const bool early_blur = h->overlap && h->prof_on != 1 && h->blur_split >= 1 && h->blur_split < nl - 1;
    bool early_done = false;
    if (h->pyr_mode == 0 && src0_aligned) {
        for (int l = 1; l < nl; ++l) {
            const OrbLevel& S = h->levels[l - 1];
            const OrbLevel& D = h->levels[l];
            const uint8_t* src = l == 1 ? d_gray : h->d_pyr + S.plane_off;
            if (h->d_rs_col8[l])
                orbk_resize_rows8(st, src, l == 1 ? stride : (size_t)S.stride, l == 1 ? frame_stride : h->pyr_frame_total, S.h,
                                  h->d_pyr + D.plane_off, D.w, D.h, (size_t)D.stride, h->pyr_frame_total, h->d_rs_col8[l], h->d_rs_row[l], nframes);
            else
                orbk_resize_rows4(st, src, l == 1 ? stride : (size_t)S.stride, l == 1 ? frame_stride : h->pyr_frame_total, S.h,
                                  h->d_pyr + D.plane_off, D.w, D.h, (size_t)D.stride, h->pyr_frame_total, h->d_rs_col[l], h->d_rs_row[l], nframes);
            if (early_blur && l == h->blur_split) {
                // the blur of the big levels 0 .. l (most of its bytes) runs on the side stream beside the rest of the chain:
                // the small levels are a few microseconds of work behind a kernel boundary each and leave the chip idle
                HIP_TRY(hipEventRecord(h->ev_mid, st));
                HIP_TRY(hipStreamWaitEvent(h->stream_b, h->ev_mid, 0));
                launch_blur(h->stream_b, 0, l + 1);
                early_done = true;
            }
        }
    } else if (!h->pyr_segs.empty()) {
        for (const slamit_orb::PyrSeg& sg : h->pyr_segs)
            orbk_pyramid(st, h->d_levels, nl, h->d_boxes + sg.box_off, h->d_tabs, sg.nregions, d_gray, stride, frame_stride, h->d_pyr,
                         sg.bufA, sg.smem, nframes, sg.first, sg.last, sg.threads);
    } else {
        for (int l = 1; l < nl; ++l) {
            const OrbLevel& S = h->levels[l - 1];
            const OrbLevel& D = h->levels[l];
            const uint8_t* src = l == 1 ? d_gray : h->d_pyr + S.plane_off;
            size_t sstride = l == 1 ? stride : (size_t)S.stride;
            size_t sframe = l == 1 ? frame_stride : h->pyr_frame_total;
            orbk_resize(st, src, S.w, S.h, sstride, sframe, h->d_pyr + D.plane_off, D.w, D.h, (size_t)D.stride,
                        h->pyr_frame_total, h->d_tab_i[l][0], h->d_tab_s[l][0], h->d_tab_i[l][1], h->d_tab_s[l][1], nframes);
        }
    }
    prof_mark(h, st, ST_RESIZE, false);
    if (h->overlap && h->prof_on != 1 && getenv("SLAMIT_ORB_FORK_EARLY")) {
        HIP_TRY(hipEventRecord(h->ev_pyr, st));
        HIP_TRY(hipStreamWaitEvent(h->stream_b, h->ev_pyr, 0));
        launch_blur(h->stream_b, 0, nl);
        HIP_TRY(hipEventRecord(h->ev_blur, h->stream_b));
    }
    // K2: FAST + NMS + per-cell threshold fallback -> candidate lists
    prof_mark(h, st, ST_FAST, true);
    orbk_fast(st, h->levels.data(), nl, h->d_cells, h->fast_cells, d_gray, stride, frame_stride, h->d_pyr, h->d_cand,
              h->cand_frame_stride, cand_count, h->p.ini_th_fast, h->p.min_th_fast, h->max_wcell, h->max_hcell, nframes);
    prof_mark(h, st, ST_FAST, false);
    // K6: blur every level.  Only the descriptor pass reads it, and it only needs the pyramid: it runs on the side stream
    // beside the octree / orientation launches (latency bound: a few hundred workgroups on 256 CUs) and joins before the
    // descriptors.  FAST (issue bound, the kernel the roofline is quoted on) keeps the chip to itself.  While every stage
    // is being timed (slamit_orb_profile(h, 1)) everything stays on one stream.
    const bool side = h->overlap && h->prof_on != 1;
    static const int fork_early = getenv("SLAMIT_ORB_FORK_EARLY") ? 1 : 0;
    if (side && !fork_early) {
        HIP_TRY(hipEventRecord(h->ev_pyr, st));
        HIP_TRY(hipStreamWaitEvent(h->stream_b, h->ev_pyr, 0));
        launch_blur(h->stream_b, early_done ? h->blur_split + 1 : 0, nl);   // the levels the early launch left
        HIP_TRY(hipEventRecord(h->ev_blur, h->stream_b));
    }
    // K4: octree
    prof_mark(h, st, ST_OCTREE, true);
    orbk_octree(st, h->d_levels, nl, h->d_cand, h->cand_frame_stride, cand_count, h->d_ws_xy, h->d_ws_node, h->d_lkp,
                h->kp_frame_stride, kp_count, h->node_cap, h->oct_key_cap, nframes, -1);
    prof_mark(h, st, ST_OCTREE, false);
    // K5: orientation
    prof_mark(h, st, ST_ANGLE, true);
    orbk_ic_angle(st, h->d_levels, nl, d_gray, stride, frame_stride, h->d_pyr, h->d_lkp, h->kp_frame_stride, kp_count,
                  h->max_kp_level, nframes);
    prof_mark(h, st, ST_ANGLE, false);
    if (side) {
        HIP_TRY(hipStreamWaitEvent(st, h->ev_blur, 0));
    } else {
        prof_mark(h, st, ST_BLUR, true);
        launch_blur(st, 0, nl);
        prof_mark(h, st, ST_BLUR, false);
    }
    // K7: descriptors + output records
    prof_mark(h, st, ST_DESCRIBE, true);
    orbk_describe(st, h->d_levels, nl, h->d_blur, h->d_lkp, h->kp_frame_stride, kp_count, d_kps, d_desc, cap, d_n_out,
                  h->max_kp_level, nframes);
    prof_mark(h, st, ST_DESCRIBE, false);
    ++h->prof_call;
    HIP_TRY(hipGetLastError());
    h->counters_clean = true;
    h->last_img0 = d_gray; h->last_stride = stride; h->last_frame = frame_stride; h->last_nframes = nframes;
    return SLAMIT_OK;
}

int slamit_orb_extract_batch(slamit_orb* h, const uint8_t* gray, size_t stride, size_t frame_stride, int nframes,
                             slamit_kp* kps, uint8_t* desc, int cap, int* n_out) {
    if (!h || !n_out || nframes < 0) return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_extract_batch: bad argument");
    if (nframes > h->p.max_batch) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_orb_extract_batch: nframes > max_batch");
    if (nframes == 0) return SLAMIT_OK;
    if (h->p.width == 0 || h->p.height == 0) {
        for (int f = 0; f < nframes; ++f) n_out[f] = 0;
        return SLAMIT_OK;
    }
    if (!gray || !kps || !desc) return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_extract_batch: null buffer");
    if (cap < h->max_out) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_orb_extract_batch: cap < slamit_orb_max_keypoints()");
    SLAMIT_USE_DEVICE(h->device);
    const int W = h->p.width, H = h->p.height;
    for (int f = 0; f < nframes; ++f)
        HIP_TRY(hipMemcpy2DAsync(h->d_in + f * h->d_in_frame, h->d_in_stride, gray + (size_t)f * frame_stride, stride, W, H,
                                 hipMemcpyHostToDevice, h->stream));
    int rc = slamit_orb_extract_batch_dev(h, h->d_in, h->d_in_stride, h->d_in_frame, nframes, h->d_out_kps, h->d_out_desc,
                                          h->max_out, h->d_out_n, h->stream);
    if (rc != SLAMIT_OK) return rc;
    // readback through the pinned block: counts, keypoints and descriptors are three copies on the stream and ONE
    // synchronisation (copying by the exact counts needs the counts on the host first, i.e. a second round trip)
    const size_t o_k = ((sizeof(int) * (size_t)nframes + 255) & ~(size_t)255), kb = sizeof(slamit_kp) * (size_t)h->max_out * nframes;
    const size_t o_d = (o_k + kb + 255) & ~(size_t)255, db = (size_t)SLAMIT_DESC_BYTES * h->max_out * nframes;
    HIP_TRY(hipMemcpyAsync(h->h_out, h->d_out_n, sizeof(int) * nframes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(h->h_out + o_k, h->d_out_kps, kb, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(h->h_out + o_d, h->d_out_desc, db, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    memcpy(n_out, h->h_out, sizeof(int) * nframes);
    for (int f = 0; f < nframes; ++f) {
        const int n = n_out[f];
        if (n > 0) {
            memcpy(kps + (size_t)f * cap, h->h_out + o_k + sizeof(slamit_kp) * (size_t)f * h->max_out, sizeof(slamit_kp) * n);
            memcpy(desc + (size_t)f * cap * SLAMIT_DESC_BYTES, h->h_out + o_d + (size_t)SLAMIT_DESC_BYTES * f * h->max_out, (size_t)SLAMIT_DESC_BYTES * n);
        }
    }
    return SLAMIT_OK;
}

int slamit_orb_extract(slamit_orb* h, const uint8_t* gray, size_t stride, slamit_kp* kps, uint8_t* desc, int cap,
                       int* n_out) {
    return slamit_orb_extract_batch(h, gray, stride, stride * (size_t)(h ? h->p.height : 0), 1, kps, desc, cap, n_out);
}

int slamit_orb_profile(slamit_orb* h, int enable, float* stage_ms, int32_t* stage_calls, int nstages) {
    if (!h) return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_profile: null handle");
    SLAMIT_USE_DEVICE(h->device);
    if (stage_ms || stage_calls) {
        for (int i = 0; i < nstages; ++i) { if (stage_ms) stage_ms[i] = 0.f; if (stage_calls) stage_calls[i] = 0; }
        for (size_t i = 0; i < h->prof_stage.size() && 2 * i + 1 < h->prof_ev.size(); ++i) {
            HIP_TRY(hipEventSynchronize(h->prof_ev[2 * i + 1]));
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, h->prof_ev[2 * i], h->prof_ev[2 * i + 1]));
            int s = h->prof_stage[i];
            if (s < nstages) { if (stage_ms) stage_ms[s] += ms; if (stage_calls) stage_calls[s] += 1; }
        }
    }
    for (hipEvent_t e : h->prof_ev) hipEventDestroy(e);
    h->prof_ev.clear(); h->prof_stage.clear();
    h->prof_on = enable;
    h->prof_call = 0;
    return SLAMIT_OK;
}

int slamit_orb_level(slamit_orb* h, int frame, int level, uint8_t* dst, size_t dst_bytes, int* w, int* h_out) {
    if (!h || level < 0 || level >= h->nlevels) return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_level: bad level");
    if (h->p.width == 0 || h->p.height == 0) return slamit_fail(SLAMIT_ERR_STATE, "slamit_orb_level: empty image");
    const OrbLevel& L = h->levels[level];
    if (w) *w = L.w;
    if (h_out) *h_out = L.h;
    if (!dst) return SLAMIT_OK;
    if (frame < 0 || frame >= h->last_nframes || !h->last_img0) return slamit_fail(SLAMIT_ERR_STATE, "slamit_orb_level: no such frame in the last extract call");
    size_t need = (size_t)(L.w + 38) * (L.h + 38);
    if (dst_bytes < need) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_orb_level: dst too small");
    SLAMIT_USE_DEVICE(h->device);
    const uint8_t* src = level == 0 ? h->last_img0 + (size_t)frame * h->last_frame
                                    : h->d_pyr + L.plane_off + (size_t)frame * h->pyr_frame_total;
    orbk_pad(h->stream, src, L.w, L.h, level == 0 ? h->last_stride : (size_t)L.stride, h->d_scratch);
    HIP_TRY(hipMemcpyAsync(dst, h->d_scratch, need, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SLAMIT_OK;
}

int slamit_orb_debug_blurred(slamit_orb* h, int frame, int level, uint8_t* dst, size_t dst_bytes, int* w, int* h_out) {
    if (!h || level < 0 || level >= h->nlevels) return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_debug_blurred: bad level");
    if (h->p.width == 0 || h->p.height == 0) return slamit_fail(SLAMIT_ERR_STATE, "slamit_orb_debug_blurred: empty image");
    const OrbLevel& L = h->levels[level];
    if (w) *w = L.w;
    if (h_out) *h_out = L.h;
    if (!dst) return SLAMIT_OK;
    if (frame < 0 || frame >= h->last_nframes) return slamit_fail(SLAMIT_ERR_STATE, "slamit_orb_debug_blurred: no such frame in the last extract call");
    if (dst_bytes < (size_t)L.w * L.h) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_orb_debug_blurred: dst too small");
    SLAMIT_USE_DEVICE(h->device);
    HIP_TRY(hipMemcpy2DAsync(dst, (size_t)L.w, h->d_blur + L.blur_off + (size_t)frame * h->blur_frame_total, (size_t)L.stride, (size_t)L.w,
                             (size_t)L.h, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return SLAMIT_OK;
}

int slamit_orb_debug_candidates(slamit_orb* h, int frame, int level, int32_t* xys, int cap, int* n_out) {
    if (!h || level < 0 || level >= h->nlevels || !n_out) return slamit_fail(SLAMIT_ERR_ARG, "slamit_orb_debug_candidates: bad argument");
    if (frame < 0 || frame >= h->last_nframes) return slamit_fail(SLAMIT_ERR_STATE, "slamit_orb_debug_candidates: no such frame");
    SLAMIT_USE_DEVICE(h->device);
    const OrbLevel& L = h->levels[level];
    int n = 0;
    HIP_TRY(hipMemcpy(&n, h->d_counts + (size_t)(frame * h->nlevels + level) * ORB_CC_PAD + 1, sizeof(int), hipMemcpyDeviceToHost));   // word 1: the count the octree pass consumed
    n = std::min(n, L.cand_cap);
    *n_out = n;
    if (!xys || n == 0) return SLAMIT_OK;
    if (cap < n) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_orb_debug_candidates: cap too small");
    unsigned long long* d_order = (unsigned long long*)h->d_scratch;
    int* d_xys = (int*)(d_order + round_up((size_t)n, 64));
    orbk_decode_candidates(h->stream, h->d_levels, level, h->d_cand + L.cand_off + (size_t)frame * h->cand_frame_stride, n, d_order, d_xys);
    std::vector<unsigned long long> order(n);
    std::vector<int> raw(3 * (size_t)n);
    HIP_TRY(hipMemcpyAsync(order.data(), d_order, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(raw.data(), d_xys, sizeof(int) * 3 * n, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    // present in the reference's vToDistributeKeys order (the device list is unordered)
    std::vector<int> idx(n);
    for (int i = 0; i < n; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int a, int b) { return order[a] < order[b]; });
    for (int i = 0; i < n; ++i) { xys[3 * i] = raw[3 * idx[i]]; xys[3 * i + 1] = raw[3 * idx[i] + 1]; xys[3 * i + 2] = raw[3 * idx[i] + 2]; }
    return SLAMIT_OK;
}

}  // extern "C"
