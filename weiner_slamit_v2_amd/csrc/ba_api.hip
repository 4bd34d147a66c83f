// ba_api.hip — C-ABI of the local bundle adjustment (include/slamit.h, slamit_ba_*).
// Host side of Optimizer::LocalBundleAdjustment (ORB_SLAM2/src/Optimizer.cc:453-778) from the
// point where the graph is assembled (:507) to the point where results are written back (:759):
// upload of the POD window, the two-stage schedule (:659-707), download.  All numerics run in
// ba_kernels.hip; there is no CPU solve path.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/slamit.h"
#include "ba_types.h"
#include "slamit_internal.h"

size_t bak_ldlt_smem(int Npad);
hipError_t bak_prepare(int Npad);
void bak_import(hipStream_t st, BaWin* wins, const double* const* in_pose, int max_kf, int nwin);
void bak_stage_begin(hipStream_t st, BaWin* wins, int nwin, int max_edge, int stage, int max_it, int robust, bool gate);
void bak_slot(hipStream_t st, BaWin* wins, int nwin, int max_kf, int max_pt, int max_edge, int Npad);
void bak_final(hipStream_t st, BaWin* wins, int nwin, int max_kf, int max_edge, double* const* out_pose,
               uint8_t* const* out_flag);

static_assert(BA_MAX_ITS == SLAMIT_BA_MAX_ITS, "stats capacity");

namespace {
inline size_t rup(size_t v, size_t a) { return (v + a - 1) / a * a; }
}

struct slamit_ba {
    int device;
    hipStream_t stream;
    int max_kf, max_pt, max_edge, max_batch;
    int Npad_max, Kpad_max, n_part;
    size_t win_bytes;          // device bytes of one window slab
    uint8_t* d_slab;           // max_batch * win_bytes
    BaWin* d_wins;             // max_batch
    BaState* d_states;         // max_batch
    double** d_out_pose;       // max_batch pointers (into the slabs)
    uint8_t** d_out_flag;
    const double** d_in_pose;
    std::vector<uint8_t> h_stage;   // pinned-like staging (plain host memory)
};

namespace {

// carve one window's arrays out of its slab; returns bytes used. `base` may be null (size query)
struct Carver {
    uint8_t* base;
    size_t off;
    template <typename T>
    T* take(size_t n) {
        off = rup(off, 256);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += sizeof(T) * n;
        return p;
    }
};

struct WinLayout {
    BaWin w;
    double* in_pose;     // 12 * n_kf staging of the caller's R|t
    double* out_pose;    // 12 * n_kf
    uint8_t* out_flag;   // n_edge
    size_t bytes;
};

WinLayout carve(uint8_t* base, int max_kf, int max_pt, int max_edge, int Npad, int Kpad, int n_part) {
    Carver c{base, 0};
    WinLayout L;
    memset(&L.w, 0, sizeof(L.w));
    BaWin& w = L.w;
    w.pose = c.take<double>(7 * (size_t)max_kf); w.pose_bak = c.take<double>(7 * (size_t)max_kf);
    w.intr = c.take<double>(4 * (size_t)max_kf); w.pose_col = c.take<int32_t>(max_kf);
    w.pt = c.take<double>(3 * (size_t)max_pt); w.pt_bak = c.take<double>(3 * (size_t)max_pt);
    w.e_kf = c.take<int32_t>(max_edge); w.e_pt = c.take<int32_t>(max_edge);
    w.e_uv = c.take<double>(2 * (size_t)max_edge); w.e_w = c.take<double>(max_edge);
    w.e_active = c.take<uint8_t>(max_edge); w.e_out1 = c.take<uint8_t>(max_edge);
    w.e_chi2 = c.take<double>(max_edge); w.e_jac = c.take<double>(21 * (size_t)max_edge);
    w.pt_ptr = c.take<int32_t>((size_t)max_pt + 1); w.pt_edges = c.take<int32_t>(max_edge);
    w.kf_ptr = c.take<int32_t>((size_t)max_kf + 1); w.kf_edges = c.take<int32_t>(max_edge);
    w.Hll = c.take<double>(6 * (size_t)max_pt); w.bl = c.take<double>(3 * (size_t)max_pt);
    w.Dinv = c.take<double>(6 * (size_t)max_pt);
    w.Hpp = c.take<double>(36 * (size_t)max_kf); w.bp = c.take<double>(6 * (size_t)max_kf + 8);
    w.GA = c.take<double>((size_t)Npad * Kpad); w.GB = c.take<double>((size_t)Npad * Kpad);
    w.part = c.take<double>((size_t)BA_SPLITS * Npad * Npad);
    w.S = c.take<double>((size_t)Npad * Npad); w.rhs = c.take<double>(Npad);
    w.x_l = c.take<double>(3 * (size_t)max_pt);
    w.chi_part = c.take<double>(n_part); w.scale_part = c.take<double>(n_part);
    L.in_pose = c.take<double>(12 * (size_t)max_kf);
    L.out_pose = c.take<double>(12 * (size_t)max_kf);
    L.out_flag = c.take<uint8_t>(max_edge);
    L.bytes = rup(c.off, 4096);
    return L;
}

}  // namespace

extern "C" {

int slamit_ba_create(int max_kf, int max_pt, int max_edge, int max_batch, int device, slamit_ba** out) {
    if (!out || max_kf < 1 || max_pt < 1 || max_edge < 1 || max_batch < 1)
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_create: bad argument");
    *out = nullptr;
    HIP_TRY(hipSetDevice(device));
    slamit_ba* h = new slamit_ba();
    h->device = device;
    h->max_kf = max_kf; h->max_pt = max_pt; h->max_edge = max_edge; h->max_batch = max_batch;
    h->Npad_max = (int)rup((size_t)6 * max_kf + 1, BA_TILE);
    h->Kpad_max = (int)rup((size_t)3 * max_pt, (size_t)BA_KC * BA_SPLITS);
    h->n_part = std::max((max_edge + 255) / 256, (std::max(8 * max_pt, max_kf) + 255) / 256) + 1;   // 8 = BA_PG lanes per point
    if (bak_ldlt_smem(h->Npad_max) > 160 * 1024 - 2048) {
        delete h;
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_create: max_kf too large for the LDS-resident LDLt panel");
    }
    WinLayout probe = carve(nullptr, max_kf, max_pt, max_edge, h->Npad_max, h->Kpad_max, h->n_part);
    h->win_bytes = probe.bytes;
    hipError_t e = hipMalloc((void**)&h->d_slab, h->win_bytes * (size_t)max_batch);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_wins, sizeof(BaWin) * max_batch);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_states, sizeof(BaState) * max_batch);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_out_pose, sizeof(double*) * max_batch);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_out_flag, sizeof(uint8_t*) * max_batch);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_in_pose, sizeof(double*) * max_batch);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = bak_prepare(h->Npad_max);
    if (e == hipSuccess) {
        std::vector<double*> op(max_batch), ip(max_batch);
        std::vector<uint8_t*> of(max_batch);
        for (int b = 0; b < max_batch; ++b) {
            WinLayout L = carve(h->d_slab + (size_t)b * h->win_bytes, max_kf, max_pt, max_edge, h->Npad_max, h->Kpad_max, h->n_part);
            op[b] = L.out_pose; of[b] = L.out_flag; ip[b] = L.in_pose;
        }
        e = hipMemcpy(h->d_out_pose, op.data(), sizeof(double*) * max_batch, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(h->d_out_flag, of.data(), sizeof(uint8_t*) * max_batch, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(h->d_in_pose, ip.data(), sizeof(double*) * max_batch, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        slamit_ba_destroy(h);
        return slamit_fail_hip(e, "slamit_ba_create");
    }
    *out = h;
    return SLAMIT_OK;
}

void slamit_ba_destroy(slamit_ba* h) {
    if (!h) return;
    hipSetDevice(h->device);
    hipFree(h->d_slab); hipFree(h->d_wins); hipFree(h->d_states); hipFree(h->d_out_pose); hipFree(h->d_out_flag);
    hipFree((void*)h->d_in_pose);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

int slamit_ba_solve_batch(slamit_ba* h, int nwin, const slamit_ba_problem* probs, const slamit_ba_opts* opts,
                          slamit_ba_result* results) {
    if (!h || !probs || !opts || !results || nwin < 0) return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_solve_batch: bad argument");
    if (nwin > h->max_batch) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_ba_solve_batch: nwin > max_batch");
    if (nwin == 0) return SLAMIT_OK;
    HIP_TRY(hipSetDevice(h->device));
    hipStream_t st = h->stream;
    // ---- validate, build CSR lists, upload ----
    std::vector<BaWin> wins(nwin);
    std::vector<WinLayout> lay(nwin);
    int mk = 1, mp = 1, me = 1, Npad = BA_TILE;
    for (int b = 0; b < nwin; ++b) {
        const slamit_ba_problem& P = probs[b];
        if (P.n_kf < 1 || P.n_pt < 0 || P.n_edge < 0 || P.n_kf > h->max_kf || P.n_pt > h->max_pt || P.n_edge > h->max_edge)
            return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_ba_solve_batch: window exceeds the handle's capacity");
        if (!P.kf_pose || !P.kf_fixed || !P.kf_intr || (P.n_pt && !P.pt_xyz) ||
            (P.n_edge && (!P.edge_kf || !P.edge_pt || !P.edge_uv || !P.edge_inv_sigma2)))
            return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_solve_batch: null input array");
        if (!results[b].kf_pose || (P.n_pt && !results[b].pt_xyz))
            return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_solve_batch: null output array");
        for (int e = 0; e < P.n_edge; ++e)
            if (P.edge_kf[e] < 0 || P.edge_kf[e] >= P.n_kf || P.edge_pt[e] < 0 || P.edge_pt[e] >= P.n_pt)
                return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_solve_batch: edge index out of range");
        mk = std::max(mk, P.n_kf); mp = std::max(mp, P.n_pt); me = std::max(me, P.n_edge);
    }
    for (int b = 0; b < nwin; ++b) {
        const slamit_ba_problem& P = probs[b];
        uint8_t* base = h->d_slab + (size_t)b * h->win_bytes;
        lay[b] = carve(base, h->max_kf, h->max_pt, h->max_edge, h->Npad_max, h->Kpad_max, h->n_part);
        BaWin& w = lay[b].w;
        w.n_kf = P.n_kf; w.n_pt = P.n_pt; w.n_edge = P.n_edge;
        std::vector<int32_t> col(P.n_kf);
        int nfree = 0;
        for (int k = 0; k < P.n_kf; ++k) col[k] = P.kf_fixed[k] ? -1 : nfree++;
        w.n_free = nfree; w.nS = 6 * nfree;
        w.Npad = (int)rup((size_t)w.nS + 1, BA_TILE);
        w.Kpad = (int)rup((size_t)std::max(3 * P.n_pt, 1), (size_t)BA_KC * BA_SPLITS);
        w.n_part = h->n_part;
        w.huber_delta = opts->huber_delta; w.chi2_gate = opts->chi2_gate;
        w.st = h->d_states + b;
        Npad = std::max(Npad, w.Npad);
        // CSR by point / by keyframe (counting sort, caller order preserved inside each list)
        std::vector<int32_t> pptr(P.n_pt + 1, 0), kptr(P.n_kf + 1, 0), pe(P.n_edge), ke(P.n_edge);
        for (int e = 0; e < P.n_edge; ++e) { ++pptr[P.edge_pt[e] + 1]; ++kptr[P.edge_kf[e] + 1]; }
        for (int p = 0; p < P.n_pt; ++p) pptr[p + 1] += pptr[p];
        for (int k = 0; k < P.n_kf; ++k) kptr[k + 1] += kptr[k];
        {
            std::vector<int32_t> pc(pptr.begin(), pptr.end() - 1), kc(kptr.begin(), kptr.end() - 1);
            for (int e = 0; e < P.n_edge; ++e) { pe[pc[P.edge_pt[e]]++] = e; ke[kc[P.edge_kf[e]]++] = e; }
        }
#define UP(dst, src, n) HIP_TRY(hipMemcpyAsync((void*)(dst), (src), sizeof(*(src)) * (size_t)(n), hipMemcpyHostToDevice, st))
        UP(lay[b].in_pose, P.kf_pose, 12 * P.n_kf);
        UP(w.intr, P.kf_intr, 4 * P.n_kf);
        UP(w.pose_col, col.data(), P.n_kf);
        if (P.n_pt) UP(w.pt, P.pt_xyz, 3 * P.n_pt);
        if (P.n_edge) {
            UP(w.e_kf, P.edge_kf, P.n_edge); UP(w.e_pt, P.edge_pt, P.n_edge);
            UP(w.e_uv, P.edge_uv, 2 * P.n_edge); UP(w.e_w, P.edge_inv_sigma2, P.n_edge);
            UP(w.pt_edges, pe.data(), P.n_edge); UP(w.kf_edges, ke.data(), P.n_edge);
        }
        UP(w.pt_ptr, pptr.data(), P.n_pt + 1); UP(w.kf_ptr, kptr.data(), P.n_kf + 1);
#undef UP
        HIP_TRY(hipMemsetAsync(w.e_active, 1, std::max(P.n_edge, 1), st));
        HIP_TRY(hipMemsetAsync(w.e_out1, 0, std::max(P.n_edge, 1), st));
        HIP_TRY(hipMemsetAsync(w.e_chi2, 0, sizeof(double) * std::max(P.n_edge, 1), st));
        HIP_TRY(hipMemsetAsync(w.st, 0, sizeof(BaState), st));
        HIP_TRY(hipStreamSynchronize(st));  // the staging vectors above go out of scope
        wins[b] = w;
    }
    HIP_TRY(hipMemcpyAsync(h->d_wins, wins.data(), sizeof(BaWin) * nwin, hipMemcpyHostToDevice, st));
    bak_import(st, h->d_wins, h->d_in_pose, mk, nwin);

    // ---- two-stage schedule (Optimizer.cc:659-707) ----
    std::vector<BaState> hs(nwin);
    bool stopped = opts->stop && *opts->stop;  // :655-657
    for (int stage = 0; stage < 2 && !stopped; ++stage) {
        const int its = stage == 0 ? opts->its_robust : opts->its_final;
        for (int b = 0; b < nwin; ++b) {  // the sparsity pattern of the operands shrinks after the gate
            HIP_TRY(hipMemsetAsync(wins[b].GA, 0, sizeof(double) * (size_t)wins[b].Npad * wins[b].Kpad, st));
            HIP_TRY(hipMemsetAsync(wins[b].GB, 0, sizeof(double) * (size_t)wins[b].Npad * wins[b].Kpad, st));
        }
        bak_stage_begin(st, h->d_wins, nwin, me, stage, its, stage == 0 ? 1 : 0, stage == 1);
        int budget = its * 10 + 1;  // at most 10 LM trials per iteration
        int chunk = std::max(its, 1);
        bool all_done = false;
        while (!all_done && budget > 0) {
            if (opts->stop && *opts->stop) { stopped = true; break; }  // SparseOptimizer::terminate()
            const int nslots = std::min(chunk, budget);
            for (int s = 0; s < nslots; ++s) bak_slot(st, h->d_wins, nwin, mk, mp, me, Npad);
            budget -= nslots;
            HIP_TRY(hipMemcpyAsync(hs.data(), h->d_states, sizeof(BaState) * nwin, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            all_done = true;
            for (int b = 0; b < nwin; ++b) all_done = all_done && hs[b].done;
            chunk = 2;
        }
        HIP_TRY(hipGetLastError());
    }
    // ---- results ----
    bak_final(st, h->d_wins, nwin, mk, me, h->d_out_pose, h->d_out_flag);
    HIP_TRY(hipMemcpyAsync(hs.data(), h->d_states, sizeof(BaState) * nwin, hipMemcpyDeviceToHost, st));
    for (int b = 0; b < nwin; ++b) {
        const slamit_ba_problem& P = probs[b];
        slamit_ba_result& R = results[b];
        HIP_TRY(hipMemcpyAsync(R.kf_pose, lay[b].out_pose, sizeof(double) * 12 * P.n_kf, hipMemcpyDeviceToHost, st));
        if (P.n_pt) HIP_TRY(hipMemcpyAsync(R.pt_xyz, wins[b].pt, sizeof(double) * 3 * P.n_pt, hipMemcpyDeviceToHost, st));
        if (P.n_edge) {
            if (R.edge_chi2) HIP_TRY(hipMemcpyAsync(R.edge_chi2, wins[b].e_chi2, sizeof(double) * P.n_edge, hipMemcpyDeviceToHost, st));
            if (R.edge_outlier) HIP_TRY(hipMemcpyAsync(R.edge_outlier, lay[b].out_flag, P.n_edge, hipMemcpyDeviceToHost, st));
            if (R.edge_stage1_outlier) HIP_TRY(hipMemcpyAsync(R.edge_stage1_outlier, wins[b].e_out1, P.n_edge, hipMemcpyDeviceToHost, st));
        }
    }
    HIP_TRY(hipStreamSynchronize(st));
    if (getenv("SLAMIT_BA_DIAG"))  // diagnostic builds only: in-kernel clock of the last LDLt launch
        fprintf(stderr, "[ba diag] ldlt shader cycles %llu, realtime ticks (100 MHz) %llu -> %.0f MHz, %.1f us\n",
                hs[0].dbg[2] - hs[0].dbg[0], hs[0].dbg[3] - hs[0].dbg[1],
                100.0 * (double)(hs[0].dbg[2] - hs[0].dbg[0]) / (double)(hs[0].dbg[3] - hs[0].dbg[1] + 1),
                (double)(hs[0].dbg[3] - hs[0].dbg[1]) / 100.0);
    if (getenv("SLAMIT_BA_DIAG"))
        fprintf(stderr, "[ba diag] ldlt phase cycles: load %llu factor %llu rows %llu writeback %llu trailing %llu backsub %llu\n",
                hs[0].dbg[4] >> 32, hs[0].dbg[4] & 0xffffffffull, hs[0].dbg[5] >> 32, hs[0].dbg[5] & 0xffffffffull,
                hs[0].dbg[6] >> 32, hs[0].dbg[6] & 0xffffffffull);
    for (int b = 0; b < nwin; ++b) {
        slamit_ba_stats* S = results[b].stats;
        if (!S) continue;
        memset(S, 0, sizeof(*S));
        for (int s = 0; s < 2; ++s) {
            S->n_its[s] = hs[b].n_its[s];
            S->chi2_init[s] = hs[b].chi2_init[s];
            for (int i = 0; i < SLAMIT_BA_MAX_ITS; ++i) { S->chi2[s][i] = hs[b].chi2[s][i]; S->lambda[s][i] = hs[b].lam[s][i]; S->trials[s][i] = hs[b].trials[s][i]; }
        }
    }
    return SLAMIT_OK;
}

int slamit_ba_solve(slamit_ba* h, const slamit_ba_problem* prob, const slamit_ba_opts* opts, slamit_ba_result* res) {
    return slamit_ba_solve_batch(h, 1, prob, opts, res);
}

}  // extern "C"
