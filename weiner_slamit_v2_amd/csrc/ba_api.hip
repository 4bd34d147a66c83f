// ba_api.hip — C-ABI of the local bundle adjustment (include/slamit.h, slamit_ba_*).
// Host side of Optimizer::LocalBundleAdjustment (ORB_SLAM2/src/Optimizer.cc:453-778) from the
// point where the graph is assembled (:507) to the point where results are written back (:759):
// upload of the POD window, the two-stage schedule (:659-707), download.  All numerics run in
// ba_kernels.hip; there is no CPU solve path.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <limits.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <exception>
#include <thread>
#include <vector>

#include "../../include/slamit.h"
#include "ba_types.h"
#include "slamit_internal.h"

size_t bak_ldlt_smem(int Npad);
hipError_t bak_prepare(int Npad);
void bak_import(hipStream_t st, BaWin* wins, const BaIo* io, int max_kf, int max_pt, int max_edge, int Npad, int nwin);
void bak_stage_begin(hipStream_t st, BaWin* wins, int nwin, int max_edge, int stage, int max_it, int robust, bool gate);
void bak_slot(hipStream_t st, BaWin* wins, int nwin, int max_kf, int max_pt, int max_edge, int Npad, bool first, unsigned solvers, hipEvent_t* ev);
int bak_solver_kind(int n, int band);
int bak_nsplit(int nwin);
void bak_final(hipStream_t st, BaWin* wins, const BaIo* io, int nwin, int max_kf, int max_pt, int max_edge);

static_assert(BA_MAX_ITS == SLAMIT_BA_MAX_ITS, "stats capacity");
static_assert(sizeof(BaState) % 8 == 0, "BaState is copied as 64-bit words");

namespace {
inline size_t rup(size_t v, size_t a) { return (v + a - 1) / a * a; }
}

struct slamit_ba {
    int device;
    hipStream_t stream;
    hipEvent_t ev[2];          // state read-backs of the LM chunks in flight
    int max_kf, max_pt, max_edge, max_batch;
    int Npad_max, Kpad_max, n_part;
    size_t win_bytes;          // device bytes of one window slab: io section (inputs | outputs) then the workspace
    size_t io_cap;             // bytes of the io section
    uint8_t* d_slab;           // max_batch * win_bytes
    BaWin* d_wins;             // max_batch
    BaState* d_states;         // max_batch
    BaIo* d_io;                // max_batch
    uint8_t* h_pin;            // pinned: every window's packed inputs, then outputs, then 2 x max_batch LM states
    size_t pin_bytes;
    // slamit_ba_profile: six timing events per LM slot of the current solve, and the phase sums of the last profiled one
    bool prof;
    std::vector<hipEvent_t> pev;
    slamit_ba_profile_out prof_last;
};

namespace {

// carve arrays out of a block; `base` may be null (size query)
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

// The io section of a window: the arrays the host writes (packed for the ACTUAL sizes of the problem, so one copy moves
// exactly what is needed) followed by the arrays it reads back.
struct IoLayout {
    // inputs
    double* in_pose; double* intr; int32_t* pose_col; double* in_pt; int32_t* e_kf; int32_t* e_pt; double* e_uv; double* e_w;
    double* e_ur; double* bf;   // stereo windows only (slamit_ba_problem::edge_ur / kf_bf)
    int32_t* pt_edges; int32_t* kf_edges; int32_t* pt_ptr; int32_t* kf_ptr;
    size_t in_bytes;
    // outputs
    size_t out_off;
    double* out_pose; double* out_pt; double* out_chi2; uint8_t* out_flag; uint8_t* out_out1; BaState* out_state;
    size_t bytes;
};

IoLayout carve_io(uint8_t* base, int n_kf, int n_pt, int n_edge, bool stereo) {
    Carver c{base, 0};
    IoLayout L;
    L.e_ur = nullptr; L.bf = nullptr;
    L.in_pose = c.take<double>(12 * (size_t)n_kf); L.intr = c.take<double>(4 * (size_t)n_kf); L.pose_col = c.take<int32_t>(n_kf);
    L.in_pt = c.take<double>(3 * (size_t)std::max(n_pt, 1));
    L.e_kf = c.take<int32_t>(std::max(n_edge, 1)); L.e_pt = c.take<int32_t>(std::max(n_edge, 1));
    L.e_uv = c.take<double>(2 * (size_t)std::max(n_edge, 1)); L.e_w = c.take<double>(std::max(n_edge, 1));
    L.pt_edges = c.take<int32_t>(std::max(n_edge, 1)); L.kf_edges = c.take<int32_t>(std::max(n_edge, 1));
    L.pt_ptr = c.take<int32_t>((size_t)n_pt + 1); L.kf_ptr = c.take<int32_t>((size_t)n_kf + 1);
    if (stereo) { L.e_ur = c.take<double>(std::max(n_edge, 1)); L.bf = c.take<double>(n_kf); }
    L.in_bytes = rup(c.off, 256);
    c.off = L.in_bytes;
    L.out_off = c.off;
    L.out_pose = c.take<double>(12 * (size_t)n_kf); L.out_pt = c.take<double>(3 * (size_t)std::max(n_pt, 1));
    L.out_chi2 = c.take<double>(std::max(n_edge, 1)); L.out_flag = c.take<uint8_t>(std::max(n_edge, 1));
    L.out_out1 = c.take<uint8_t>(std::max(n_edge, 1)); L.out_state = c.take<BaState>(1);
    L.bytes = rup(c.off, 4096);
    return L;
}

// the workspace of a window (sized for the handle's maxima); returns bytes used
size_t carve_work(uint8_t* base, BaWin& w, int max_kf, int max_pt, int max_edge, int Npad, int Kpad, int n_part) {
    Carver c{base, 0};
    w.pose = c.take<double>(7 * (size_t)max_kf); w.pose_bak = c.take<double>(7 * (size_t)max_kf);
    w.pt = c.take<double>(3 * (size_t)max_pt); w.pt_bak = c.take<double>(3 * (size_t)max_pt);
    w.e_active = c.take<uint8_t>(max_edge); w.e_out1 = c.take<uint8_t>(max_edge);
    w.e_chi2 = c.take<double>(max_edge); w.e_jac = c.take<double>(BA_JAC_STEREO * (size_t)max_edge);
    w.Hll = c.take<double>(6 * (size_t)max_pt); w.bl = c.take<double>(3 * (size_t)max_pt);
    w.Dinv = c.take<double>(6 * (size_t)max_pt);
    w.Hpp = c.take<double>(36 * (size_t)max_kf); w.bp = c.take<double>(6 * (size_t)max_kf + 8);
    w.GA = c.take<double>((size_t)Npad * Kpad);
    w.part = c.take<double>((size_t)BA_SPLITS * Npad * Npad);
    w.S = c.take<double>((size_t)Npad * Npad); w.Sb = c.take<double>(((size_t)Npad + 1) * 64); w.rhs = c.take<double>(Npad);
    w.x_l = c.take<double>(3 * (size_t)max_pt);
    w.chi_part = c.take<double>(n_part); w.scale_part = c.take<double>(n_part);
    return rup(c.off, 4096);
}

}  // namespace

extern "C" {

int slamit_ba_create(int max_kf, int max_pt, int max_edge, int max_batch, int device, slamit_ba** out) {
    if (!out || max_kf < 1 || max_pt < 1 || max_edge < 1 || max_batch < 1)
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_create: bad argument");
    *out = nullptr;
    SLAMIT_USE_DEVICE(device);
    slamit_ba* h = new slamit_ba();
    h->device = device;
    h->max_kf = max_kf; h->max_pt = max_pt; h->max_edge = max_edge; h->max_batch = max_batch;
    h->Npad_max = (int)rup((size_t)6 * max_kf + 1, BA_TILE);
    h->Kpad_max = (int)rup((size_t)3 * max_pt, (size_t)BA_KC * BA_SPLITS);
    h->n_part = std::max((max_edge + 255) / 256, (std::max(8 * max_pt, max_kf) + 255) / 256) + 1;   // 8 = BA_PG lanes per point
    if (bak_ldlt_smem(h->Npad_max) > 160 * 1024 - 2048 || h->Npad_max > BA_TILE * BA_MAX_TILES || h->Npad_max > 32 * BA_MAX_PANELS) {
        delete h;
        return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_create: max_kf too large for the LDS-resident LDLt panel");
    }
    {
        BaWin probe;
        h->io_cap = carve_io(nullptr, max_kf, max_pt, max_edge, true).bytes;
        h->win_bytes = h->io_cap + carve_work(nullptr, probe, max_kf, max_pt, max_edge, h->Npad_max, h->Kpad_max, h->n_part);
    }
    hipError_t e = hipMalloc((void**)&h->d_slab, h->win_bytes * (size_t)max_batch);
    // one block: the LM states in REVERSE order right in front of the window table (state b = (BaState*)d_wins - (b + 1)): a kernel
    // finds its state from the table's address alone (BA_ST, ba_kernels.hip)
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_states, (sizeof(BaState) + sizeof(BaWin)) * (size_t)max_batch);
    if (e == hipSuccess) h->d_wins = reinterpret_cast<BaWin*>(h->d_states + max_batch);
    if (e == hipSuccess) e = hipMalloc((void**)&h->d_io, sizeof(BaIo) * max_batch);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&h->ev[i], hipEventDisableTiming);
    if (e == hipSuccess) e = bak_prepare(h->Npad_max);
    if (e != hipSuccess) {
        slamit_ba_destroy(h);
        return slamit_fail_hip(e, "slamit_ba_create");
    }
    *out = h;
    return SLAMIT_OK;
}

void slamit_ba_destroy(slamit_ba* h) {
    if (!h) return;
    SlamitDeviceGuard guard(h->device);
    hipFree(h->d_slab); hipFree(h->d_states); hipFree(h->d_io);
    if (h->h_pin) hipHostFree(h->h_pin);
    for (int i = 0; i < 2; ++i) if (h->ev[i]) hipEventDestroy(h->ev[i]);
    for (hipEvent_t e : h->pev) hipEventDestroy(e);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
}

static int ba_solve_batch_impl(slamit_ba* h, int nwin, const slamit_ba_problem* probs, const slamit_ba_opts* opts, slamit_ba_result* results);

int slamit_ba_solve_batch(slamit_ba* h, int nwin, const slamit_ba_problem* probs, const slamit_ba_opts* opts,
                          slamit_ba_result* results) {
    try {
        return ba_solve_batch_impl(h, nwin, probs, opts, results);
    } catch (const std::exception& e) {   // (host containers: std::bad_alloc, std::length_error)
        return slamit_fail(SLAMIT_ERR_DEVICE, e.what());
    } catch (...) {
        return slamit_fail(SLAMIT_ERR_DEVICE, "slamit_ba_solve_batch: unexpected exception");
    }
}

// Column order of the free keyframes in the reduced system.  The banded solve and the floating-window Schur product want keyframes that share
// points to be NEIGHBOURS in that order; a caller that lists its local window by co-visibility weight (Optimizer.cc:456-470 walks
// GetVectorCovisibleKeyFrames) instead of along the trajectory gives the same graph in a scattered order.  If a reverse Cuthill-McKee
// order of the co-visibility graph (free keyframes; an edge = a shared point) has a narrower band than the caller's, `col` is renumbered
// to it; g2o orders the same system by approximate minimum degree (linear_solver_eigen.h:77-92) -- any order gives the same solution up
// to rounding.  The caller's order is kept when the banded solve takes it as it is, or when no order is narrower (SLAMIT_BA_KEEP_ORDER=1: always).
// `span` = the widest point of the caller's order (last - first column it is seen from), `complete` = some point is seen from every
// free keyframe (the graph is complete: no order is narrower): both come out of the pass over the edges the caller makes anyway, and
// decide without one of their own -- an order whose band the banded solve already takes (<= 9 keyframes) is kept as it is.
static bool ba_order_columns(const slamit_ba_problem& P, int32_t* col, int nfree, int span, bool complete) {
    if (nfree < 3 || complete || 6 * span + 5 <= 59 || getenv("SLAMIT_BA_KEEP_ORDER")) return false;
    const int W64 = (nfree + 63) / 64;
    std::vector<uint64_t> adj((size_t)nfree * W64, 0), seen((size_t)std::max(P.n_pt, 1) * W64, 0);
    for (int e = 0; e < P.n_edge; ++e) {
        const int c = col[P.edge_kf[e]];
        if (c >= 0) seen[(size_t)P.edge_pt[e] * W64 + (c >> 6)] |= 1ull << (c & 63);
    }
    for (int p = 0; p < P.n_pt; ++p) {
        const uint64_t* m = &seen[(size_t)p * W64];
        for (int w = 0; w < W64; ++w)
            for (uint64_t bits = m[w]; bits; bits &= bits - 1) {
                const int c = 64 * w + __builtin_ctzll(bits);
                for (int v = 0; v < W64; ++v) adj[(size_t)c * W64 + v] |= m[v];
            }
    }
    auto has = [&](int a, int b2) { return (adj[(size_t)a * W64 + (b2 >> 6)] >> (b2 & 63)) & 1ull; };
    auto band_of = [&](const std::vector<int>& pos) {   // max over columns of (position - leftmost coupled position), in keyframes
        int band = 0;
        for (int a = 0; a < nfree; ++a)
            for (int b2 = 0; b2 < nfree; ++b2)
                if (a != b2 && has(a, b2)) band = std::max(band, pos[a] - pos[b2]);
        return band;
    };
    std::vector<int> ident(nfree), deg(nfree, 0);
    for (int a = 0; a < nfree; ++a) {
        ident[a] = a;
        for (int w = 0; w < W64; ++w) deg[a] += __builtin_popcountll(adj[(size_t)a * W64 + w]);
    }
    const int band0 = band_of(ident);
    if (band0 <= 1) return false;
    // Cuthill-McKee per component from a node of minimum degree, neighbours by increasing degree (ties: the caller's order), then reversed
    std::vector<int> order; order.reserve(nfree);
    std::vector<char> used(nfree, 0);
    while ((int)order.size() < nfree) {
        int start = -1;
        for (int a = 0; a < nfree; ++a) if (!used[a] && (start < 0 || deg[a] < deg[start])) start = a;
        size_t head = order.size();
        order.push_back(start); used[start] = 1;
        while (head < order.size()) {
            const int a = order[head++];
            const size_t first = order.size();
            for (int b2 = 0; b2 < nfree; ++b2) if (!used[b2] && has(a, b2)) { order.push_back(b2); used[b2] = 1; }
            std::stable_sort(order.begin() + first, order.end(), [&](int x, int y) { return deg[x] < deg[y]; });
        }
    }
    std::reverse(order.begin(), order.end());
    std::vector<int> pos(nfree);
    for (int i = 0; i < nfree; ++i) pos[order[i]] = i;
    if (band_of(pos) >= band0) return false;
    for (int k = 0; k < P.n_kf; ++k) if (col[k] >= 0) col[k] = pos[col[k]];
    return true;
}

static int ba_solve_batch_impl(slamit_ba* h, int nwin, const slamit_ba_problem* probs, const slamit_ba_opts* opts,
                               slamit_ba_result* results) {
    if (!h || !probs || !opts || !results || nwin < 0) return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_solve_batch: bad argument");
    if (nwin > h->max_batch) return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_ba_solve_batch: nwin > max_batch");
    if (nwin == 0) return SLAMIT_OK;
    SLAMIT_USE_DEVICE(h->device);
    hipStream_t st = h->stream;
    static const bool timing = getenv("SLAMIT_BA_TIMING") != nullptr;   // diagnostic: host phases of the call on stderr
    const auto tclk = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_in = tclk();
    double t_val = 0, t_prep = 0, t_queue = 0, t_loop = 0;
    // ---- validate ----
    int mk = 1, mp = 1, me = 1, Npad = BA_TILE;
    for (int b = 0; b < nwin; ++b) {
        const slamit_ba_problem& P = probs[b];
        if (P.n_kf < 1 || P.n_pt < 0 || P.n_edge < 0 || P.n_kf > h->max_kf || P.n_pt > h->max_pt || P.n_edge > h->max_edge)
            return slamit_fail(SLAMIT_ERR_CAPACITY, "slamit_ba_solve_batch: window exceeds the handle's capacity");
        if (!P.kf_pose || !P.kf_fixed || !P.kf_intr || (P.n_pt && !P.pt_xyz) ||
            (P.n_edge && (!P.edge_kf || !P.edge_pt || !P.edge_uv || !P.edge_inv_sigma2)))
            return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_solve_batch: null input array");
        if (P.edge_ur && !P.kf_bf)
            return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_solve_batch: stereo observations (edge_ur) without the keyframes' bf (kf_bf)");
        if (!results[b].kf_pose || (P.n_pt && !results[b].pt_xyz))
            return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_solve_batch: null output array");
        mk = std::max(mk, P.n_kf); mp = std::max(mp, P.n_pt); me = std::max(me, P.n_edge);
    }
    // (the edges' indices are checked by the threads that prepare the windows, before anything is packed)
    t_val = tclk();
    // ---- one pinned block: [inputs of window 0 | inputs of window 1 | ...][outputs ...][2 x nwin LM states] ----
    std::vector<size_t> in_off(nwin), out_off(nwin);
    std::vector<IoLayout> dio(nwin);   // device addresses inside the slabs
    size_t pin_need = 0;
    for (int b = 0; b < nwin; ++b) {
        dio[b] = carve_io(h->d_slab + (size_t)b * h->win_bytes, probs[b].n_kf, probs[b].n_pt, probs[b].n_edge, probs[b].edge_ur != nullptr);
        in_off[b] = pin_need; pin_need += dio[b].in_bytes;
    }
    for (int b = 0; b < nwin; ++b) { out_off[b] = pin_need; pin_need += dio[b].bytes - dio[b].out_off; }
    const size_t st_off = rup(pin_need, 256);
    pin_need = st_off + 2 * sizeof(BaState) * (size_t)nwin;
    if (pin_need > h->pin_bytes) {
        HIP_TRY(hipStreamSynchronize(st));
        if (h->h_pin) hipHostFree(h->h_pin);
        h->h_pin = nullptr; h->pin_bytes = 0;
        HIP_TRY(hipHostMalloc((void**)&h->h_pin, pin_need + pin_need / 4, hipHostMallocDefault));
        h->pin_bytes = pin_need + pin_need / 4;
    }
    std::vector<BaWin> wins(nwin);
    std::vector<BaIo> io(nwin);
    std::vector<std::vector<int32_t> > perm(nwin);   // device point index -> caller's point index
    std::vector<double> exec_mflop(nwin, 0.0);       // what the Schur product multiplies per trial (slamit_ba_profile_out)
    // per-window preparation (structure, CSR lists, packing into the pinned block) is host work of ~10 ns per edge: windows
    // are independent, so a batch is prepared by a few host threads; the copies are queued afterwards, in order
    std::atomic<bool> bad_index(false);
    auto prepare = [&](int b) {
        const slamit_ba_problem& P = probs[b];
        for (int e = 0; e < P.n_edge; ++e)
            if (P.edge_kf[e] < 0 || P.edge_kf[e] >= P.n_kf || P.edge_pt[e] < 0 || P.edge_pt[e] >= P.n_pt) { bad_index = true; return; }
        uint8_t* slab = h->d_slab + (size_t)b * h->win_bytes;
        (void)slab;
        BaWin& w = wins[b];
        memset(&w, 0, sizeof(w));
        carve_work(slab + h->io_cap, w, h->max_kf, h->max_pt, h->max_edge, h->Npad_max, h->Kpad_max, h->n_part);
        const IoLayout& D = dio[b];
        const IoLayout H = carve_io(h->h_pin + in_off[b], P.n_kf, P.n_pt, P.n_edge, P.edge_ur != nullptr);   // the same packing in the pinned block
        w.intr = D.intr; w.pose_col = D.pose_col; w.e_kf = D.e_kf; w.e_pt = D.e_pt; w.e_uv = D.e_uv; w.e_w = D.e_w;
        w.pt_edges = D.pt_edges; w.kf_edges = D.kf_edges; w.pt_ptr = D.pt_ptr; w.kf_ptr = D.kf_ptr;
        io[b].in_pose = D.in_pose; io[b].in_pt = D.in_pt; io[b].out_pose = D.out_pose; io[b].out_pt = D.out_pt;
        io[b].out_chi2 = D.out_chi2; io[b].out_flag = D.out_flag; io[b].out_out1 = D.out_out1; io[b].out_state = D.out_state;
        w.n_kf = P.n_kf; w.n_pt = P.n_pt; w.n_edge = P.n_edge;
        int32_t* col = H.pose_col;
        int nfree = 0;
        for (int k = 0; k < P.n_kf; ++k) col[k] = P.kf_fixed[k] ? -1 : nfree++;
        w.n_free = nfree; w.nS = 6 * nfree;
        w.Npad = (int)rup((size_t)w.nS + 1, BA_TILE);
        w.Kpad = (int)rup((size_t)std::max(3 * P.n_pt, 1), (size_t)BA_KC * BA_SPLITS);
        w.n_part = h->n_part;
        w.huber_delta = opts->huber_delta; w.chi2_gate = opts->chi2_gate;
        // stereo observations (EdgeStereoSE3ProjectXYZ): three residual rows per edge in this window, own Huber width and gate
        w.huber_delta_s = opts->huber_delta_stereo > 0 ? opts->huber_delta_stereo : (double)(float)sqrt(7.815);   // Optimizer.cc:570
        w.chi2_gate_s = opts->chi2_gate_stereo > 0 ? opts->chi2_gate_stereo : 7.815;                           // Optimizer.cc:696, 740
        w.nrow = P.edge_ur ? 3 : 2;
        w.e_ur = D.e_ur; w.bf = D.bf;
        w.st = reinterpret_cast<BaState*>(h->d_wins) - (b + 1);
        // ---- structure of the window (g2o's BlockSolver / SimplicialLDLT exploit the same sparsity on the CPU,
        // block_solver.hpp:381-432, linear_solver_eigen.h:94-124) ----
        // Points are stored on the device sorted by the first free keyframe that observes them: the rows of the Schur operand GA that
        // belong to a 64-row tile then have their non-zeros in one k range, and the Schur product skips the rest.
        std::vector<int32_t> minc(P.n_pt, INT32_MAX), maxc(P.n_pt, -1);
        {
            std::vector<int32_t> seen_by(P.n_pt, 0);
            for (int e = 0; e < P.n_edge; ++e) {
                const int c = col[P.edge_kf[e]], p2 = P.edge_pt[e];
                if (c >= 0) { minc[p2] = std::min(minc[p2], c); maxc[p2] = std::max(maxc[p2], c); ++seen_by[p2]; }
            }
            int span = 0;
            bool complete = false;
            for (int p2 = 0; p2 < P.n_pt; ++p2) {
                if (maxc[p2] >= 0) span = std::max(span, maxc[p2] - minc[p2]);
                complete = complete || (maxc[p2] - minc[p2] + 1 == nfree && seen_by[p2] >= nfree);
            }
            if (ba_order_columns(P, col, nfree, span, complete)) {   // renumbered: the points' column ranges once more
                std::fill(minc.begin(), minc.end(), INT32_MAX); std::fill(maxc.begin(), maxc.end(), -1);
                for (int e = 0; e < P.n_edge; ++e) {
                    const int c = col[P.edge_kf[e]], p2 = P.edge_pt[e];
                    if (c >= 0) { minc[p2] = std::min(minc[p2], c); maxc[p2] = std::max(maxc[p2], c); }
                }
            }
        }
        std::vector<int32_t>& new2old = perm[b];
        new2old.resize(P.n_pt);
        std::vector<int32_t> old2new(P.n_pt);
        for (int p2 = 0; p2 < P.n_pt; ++p2) new2old[p2] = p2;
        std::stable_sort(new2old.begin(), new2old.end(), [&](int a, int b2) { return minc[a] != minc[b2] ? minc[a] < minc[b2] : maxc[a] < maxc[b2]; });
        for (int p2 = 0; p2 < P.n_pt; ++p2) old2new[new2old[p2]] = p2;
        {
            // per pose column: range of (sorted) points it observes, and the first column it is coupled with
            std::vector<int32_t> plo(std::max(nfree, 1), INT32_MAX), phi(std::max(nfree, 1), -1), fcol(std::max(nfree, 1));
            for (int c = 0; c < nfree; ++c) fcol[c] = c;
            for (int e = 0; e < P.n_edge; ++e) {
                const int c = col[P.edge_kf[e]], po = P.edge_pt[e];
                if (c < 0) continue;
                const int pn = old2new[po];
                plo[c] = std::min(plo[c], pn); phi[c] = std::max(phi[c], pn);
                fcol[c] = std::min(fcol[c], minc[po]);
            }
            const int T = w.Npad / BA_TILE, kmax = w.Kpad;
            for (int t = 0; t < T; ++t) {
                int lo = INT32_MAX, hi = -1;
                for (int c = 0; c < nfree; ++c) {
                    if (6 * c + 5 < BA_TILE * t || 6 * c >= BA_TILE * (t + 1) || phi[c] < 0) continue;   // pose rows outside the tile / no points
                    lo = std::min(lo, 3 * plo[c]); hi = std::max(hi, 3 * phi[c] + 3);
                }
                if (hi < 0) { lo = 0; hi = 0; }
                lo = lo / BA_KC * BA_KC; hi = std::min((hi + BA_KC - 1) / BA_KC * BA_KC, kmax);
                w.tile_alo[t] = lo; w.tile_ahi[t] = hi; w.tile_blo[t] = lo; w.tile_bhi[t] = hi;
                if (w.nS >= BA_TILE * t && w.nS < BA_TILE * (t + 1)) { w.tile_blo[t] = 0; w.tile_bhi[t] = kmax; }   // row nS = the right-hand side's row: every point
            }
            // LDLt: row envelope.  first[r] = 6 * fcol[r / 6]; panel i (columns 32 i ..) only touches rows r with first[r] < 32 i + 32
            const int n = w.nS;
            for (int i = 0; i < BA_MAX_PANELS; ++i) { w.panel_hi[i] = (int16_t)std::max(n - 1, 0); w.back_lo[i] = 0; }
            for (int i = 0; 32 * i < n; ++i) {
                const int jb = 32 * i, pend = std::min(jb + 32, n);
                int hi = pend - 1, lo = jb;
                for (int c = 0; c < nfree; ++c) {
                    if (6 * fcol[c] < pend) hi = std::max(hi, 6 * c + 5);                               // row block c reaches into the panel's columns
                    if (6 * c + 5 >= jb && 6 * c < pend) lo = std::min(lo, 6 * fcol[c]);               // rows of the panel: leftmost column
                }
                w.panel_hi[i] = (int16_t)std::min(hi, n - 1);
                w.back_lo[i] = (int16_t)lo;
            }
            int band = 0;   // a window whose keyframes only share points with their neighbours has a narrow band: LDLt inside LDS
            for (int c = 0; c < nfree; ++c) band = std::max(band, 6 * c + 5 - 6 * fcol[c]);
            w.band = std::min(band, std::max(n - 1, 0));
            w.solver = bak_solver_kind(n, w.band);
            // ---- the Schur product over floating row windows (BaWin::sf_*), when every k slab's rows fit one ----
            w.sf_groups = 0;
            const char* sf_env = getenv("SLAMIT_BA_SF");        // 0: always the tiled product (A/B runs and the tests that compare the two)
            const char* cap_env = getenv("SLAMIT_BA_SF_CAP");   // slabs per group
            if (!(sf_env && atoi(sf_env) == 0) && nfree > 0) {
                const int nslab_all = w.Kpad / BA_KC;
                std::vector<int32_t> slo(nslab_all, INT32_MAX), shi(nslab_all, -1);
                for (int pn = 0; pn < P.n_pt; ++pn) {
                    const int po = new2old[pn];
                    if (maxc[po] < 0) continue;   // no free keyframe observes it: its columns stay zero
                    for (int j = 0; j < 3; ++j) {
                        const int sl = (3 * pn + j) / BA_KC;
                        slo[sl] = std::min(slo[sl], 6 * minc[po]); shi[sl] = std::max(shi[sl], 6 * maxc[po] + 5);
                    }
                }
                // a single window wants many short workgroups (latency), a batch fewer partial tiles to write and to add
                const int cap = cap_env && atoi(cap_env) > 0 ? atoi(cap_env) : nwin >= 16 ? 8 : 4;
                const int maxg = (int)std::min<size_t>(std::min<size_t>(BA_SF_MAXG, (size_t)BA_SPLITS * h->Npad_max * h->Npad_max / (BA_TILE * BA_TILE)),   // what `part` holds
                                                       (size_t)(T * (T + 1) / 2) * bak_nsplit(nwin));                                          // workgroups of the launch
                bool ok = true;
                int G = 0, cnt = 0, clo = 0, chi = 0, kstart = 0;
                auto close = [&](int kend) {
                    if (!cnt) return;
                    if (G == maxg) { ok = false; return; }
                    w.sf_row[G] = (int16_t)clo; w.sf_k0[G] = (int16_t)kstart; w.sf_k1[G] = (int16_t)kend;
                    ++G; cnt = 0;
                };
                for (int sl = 0; sl < nslab_all && ok; ++sl) {
                    if (shi[sl] < 0) { close(sl); continue; }
                    if (shi[sl] - slo[sl] + 1 > BA_SF_ROWS || nslab_all > INT16_MAX) { ok = false; break; }
                    if (cnt && (std::max(chi, shi[sl]) - std::min(clo, slo[sl]) + 1 > BA_SF_ROWS || cnt == cap)) close(sl);
                    if (!cnt) { clo = slo[sl]; chi = shi[sl]; kstart = sl; }
                    else { clo = std::min(clo, slo[sl]); chi = std::max(chi, shi[sl]); }
                    ++cnt;
                }
                if (ok) close(nslab_all);
                for (int g = 1; g < G && ok; ++g) if (w.sf_row[g] < w.sf_row[g - 1]) ok = false;   // (sorted points: cannot happen)
                if (ok && G > 0) {
                    w.sf_groups = G;
                    int ga = 0, gb = -1;   // groups whose window reaches row r (first) / has begun at row r (last)
                    for (int r = 0; r < w.Npad; ++r) {
                        while (ga < G && w.sf_row[ga] + BA_SF_ROWS - 1 < r) ++ga;
                        while (gb + 1 < G && w.sf_row[gb + 1] <= r) ++gb;
                        w.sf_glo[r] = (int16_t)ga; w.sf_ghi[r] = (int16_t)gb;
                    }
                    // what k_zero_operands clears once per solve has to cover what the windows read
                    for (int g = 0; g < G; ++g) {
                        const int t0 = w.sf_row[g] / BA_TILE, t1 = std::min(w.sf_row[g] + BA_SF_ROWS - 1, w.Npad - 1) / BA_TILE;
                        for (int t = t0; t <= t1; ++t) {
                            const int lo = w.sf_k0[g] * BA_KC, hi = w.sf_k1[g] * BA_KC;
                            if (w.tile_ahi[t] <= w.tile_alo[t]) { w.tile_alo[t] = lo; w.tile_ahi[t] = hi; }
                            else { w.tile_alo[t] = std::min(w.tile_alo[t], lo); w.tile_ahi[t] = std::max(w.tile_ahi[t], hi); }
                            if (w.tile_bhi[t] <= w.tile_blo[t]) { w.tile_blo[t] = lo; w.tile_bhi[t] = hi; }
                            else { w.tile_blo[t] = std::min(w.tile_blo[t], lo); w.tile_bhi[t] = std::max(w.tile_bhi[t], hi); }
                        }
                        exec_mflop[b] += 2.0 * BA_TILE * BA_TILE * BA_KC * (w.sf_k1[g] - w.sf_k0[g]) * 1e-6;
                    }
                }
            }
            if (!w.sf_groups)
            for (int I = 0; I < T; ++I)   // the product's granule: 64 x 64 tile pairs (I <= J) over the k range both have non-zeros in,
                for (int J = I; J < T; ++J) {   // less the tiles a banded window's solver never reads (schur_tile_needed, ba_kernels.hip)
                    if (w.solver == BA_SOLVER_BAND && BA_TILE * J - (BA_TILE * I + BA_TILE - 1) > w.band && w.nS / BA_TILE != J) continue;
                    exec_mflop[b] += 2.0 * BA_TILE * BA_TILE * std::max(0, std::min(w.tile_ahi[I], w.tile_bhi[J]) - std::max(w.tile_alo[I], w.tile_blo[J])) * 1e-6;
                }
        }
        // ---- inputs, straight into the pinned block: CSR by point / by keyframe (counting sort, caller order kept
        // inside each list), points in device order ----
        memcpy(H.in_pose, P.kf_pose, sizeof(double) * 12 * (size_t)P.n_kf);
        memcpy(H.intr, P.kf_intr, sizeof(double) * 4 * (size_t)P.n_kf);
        for (int p2 = 0; p2 < P.n_pt; ++p2) for (int j = 0; j < 3; ++j) H.in_pt[3 * (size_t)p2 + j] = P.pt_xyz[3 * (size_t)new2old[p2] + j];
        if (P.n_edge) {
            memcpy(H.e_kf, P.edge_kf, sizeof(int32_t) * (size_t)P.n_edge);
            memcpy(H.e_uv, P.edge_uv, sizeof(double) * 2 * (size_t)P.n_edge);
            memcpy(H.e_w, P.edge_inv_sigma2, sizeof(double) * (size_t)P.n_edge);
            if (P.edge_ur) memcpy(H.e_ur, P.edge_ur, sizeof(double) * (size_t)P.n_edge);
        }
        if (P.edge_ur) memcpy(H.bf, P.kf_bf, sizeof(double) * (size_t)P.n_kf);
        int32_t* pptr = H.pt_ptr; int32_t* kptr = H.kf_ptr;
        for (int p2 = 0; p2 <= P.n_pt; ++p2) pptr[p2] = 0;
        for (int k = 0; k <= P.n_kf; ++k) kptr[k] = 0;
        for (int e = 0; e < P.n_edge; ++e) { H.e_pt[e] = old2new[P.edge_pt[e]]; ++pptr[H.e_pt[e] + 1]; ++kptr[P.edge_kf[e] + 1]; }
        for (int p2 = 0; p2 < P.n_pt; ++p2) pptr[p2 + 1] += pptr[p2];
        for (int k = 0; k < P.n_kf; ++k) kptr[k + 1] += kptr[k];
        {
            std::vector<int32_t> pc(pptr, pptr + P.n_pt), kc(kptr, kptr + P.n_kf);
            for (int e = 0; e < P.n_edge; ++e) { H.pt_edges[pc[H.e_pt[e]]++] = e; H.kf_edges[kc[P.edge_kf[e]]++] = e; }
        }
    };
    {
        // nothing thrown in here may cross the C boundary: a worker records a failed preparation (bad_alloc in its vectors), a thread
        // that cannot be created leaves its share to the threads that exist and to the caller's own
        const int nthreads = std::max(1, std::min(std::min(nwin, 16), (int)std::thread::hardware_concurrency()));
        std::atomic<int> next(0);
        std::atomic<bool> failed(false);
        auto work = [&]() {
            for (int b = next.fetch_add(1); b < nwin; b = next.fetch_add(1)) {
                try { prepare(b); } catch (...) { failed = true; }
            }
        };
        std::vector<std::thread> pool;
        try {
            for (int t = 1; t < nthreads; ++t) pool.emplace_back(work);
        } catch (...) {}
        work();
        for (std::thread& t : pool) t.join();
        if (failed) return slamit_fail(SLAMIT_ERR_DEVICE, "slamit_ba_solve_batch: out of host memory while preparing the windows");
        if (bad_index) return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_solve_batch: edge index out of range");
    }
    t_prep = tclk();
    unsigned solvers = 0;
    for (int b = 0; b < nwin; ++b) {
        Npad = std::max(Npad, wins[b].Npad);
        solvers |= 1u << wins[b].solver;
        HIP_TRY(hipMemcpyAsync(h->d_slab + (size_t)b * h->win_bytes, h->h_pin + in_off[b], dio[b].in_bytes, hipMemcpyHostToDevice, st));
    }
    HIP_TRY(hipMemcpyAsync(h->d_wins, wins.data(), sizeof(BaWin) * nwin, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->d_io, io.data(), sizeof(BaIo) * nwin, hipMemcpyHostToDevice, st));
    bak_import(st, h->d_wins, h->d_io, mk, mp, me, Npad, nwin);   // also zeroes the Schur operands' ranges and the LM states

    // ---- two-stage schedule (Optimizer.cc:659-707) ----
    // LM trial slots are enqueued in chunks; the windows' states come back through the pinned block one chunk LATE (the
    // next chunk is already queued when the host looks at the previous one: no bubble between chunks), and *stop is polled
    // before every chunk like SparseOptimizer::terminate() (sparse_optimizer.cpp:376) is before every iteration.  Slots of
    // a finished stage return at once, but a slot queued in vain still costs its eleven launches (~30 us): the first chunk
    // of a stage is the number of trials the stage cannot do without (one per iteration in the robust stage; three in the
    // final one, whose "no progress three times" rule can end it that early), the chunks after it are single slots.
    // (Chunks of two throughout queued 20 slots for the 14 trials of a window-8 solve.)
    t_queue = tclk();
    BaState* hs[2] = {reinterpret_cast<BaState*>(h->h_pin + st_off), reinterpret_cast<BaState*>(h->h_pin + st_off) + nwin};
    bool stopped = opts->stop && *opts->stop;  // :655-657
    size_t pev_used = 0;   // profiling solves: six events per queued slot
    auto slot_events = [&]() -> hipEvent_t* {
        if (!h->prof) return nullptr;
        while (h->pev.size() < pev_used + 6) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            h->pev.push_back(e);
        }
        pev_used += 6;
        return h->pev.data() + pev_used - 6;
    };
    for (int stage = 0; stage < 2 && !stopped; ++stage) {
        const int its = stage == 0 ? opts->its_robust : opts->its_final;
        bak_stage_begin(st, h->d_wins, nwin, me, stage, its, stage == 0 ? 1 : 0, stage == 1);
        int budget = its * 10 + 1;  // at most 10 LM trials per iteration
        int cur = 0, pending = -1;
        bool all_done = false, first = true;
        static const int chunk_env = getenv("SLAMIT_BA_CHUNK") ? atoi(getenv("SLAMIT_BA_CHUNK")) : 0;   // > 0: fixed chunk size (A/B runs)
        while (!all_done) {
            if (opts->stop && *opts->stop) { stopped = true; break; }
            if (budget > 0) {
                const int want = chunk_env > 0 ? chunk_env : first ? std::max(1, std::min(stage == 0 ? its : 3, std::min(its, 4))) : 1;
                const int nslots = std::min(want, budget);
                const bool was_first = first;
                first = false;
                static const bool no_fuse = getenv("SLAMIT_BA_NO_FUSE") && atoi(getenv("SLAMIT_BA_NO_FUSE"));   // A/B runs: every slot as the first
                for (int sl = 0; sl < nslots; ++sl) bak_slot(st, h->d_wins, nwin, mk, mp, me, Npad, (was_first && sl == 0) || no_fuse, solvers, slot_events());
                budget -= nslots;
                HIP_TRY(hipMemcpyAsync(hs[cur], reinterpret_cast<BaState*>(h->d_wins) - nwin, sizeof(BaState) * nwin, hipMemcpyDeviceToHost, st));   // (reverse order: only `done` of all is read)
                HIP_TRY(hipEventRecord(h->ev[cur], st));
            }
            if (pending >= 0) {
                HIP_TRY(hipEventSynchronize(h->ev[pending]));
                all_done = true;
                for (int b = 0; b < nwin; ++b) all_done = all_done && hs[pending][b].done;
            }
            if (budget <= 0 && pending == cur) break;   // nothing new was queued: the last read-back has been looked at
            pending = cur;
            if (budget > 0) cur ^= 1;
        }
        HIP_TRY(hipGetLastError());
    }
    t_loop = tclk();
    // ---- results: one copy per window out of its output section ----
    bak_final(st, h->d_wins, h->d_io, nwin, mk, mp, me);
    for (int b = 0; b < nwin; ++b)
        HIP_TRY(hipMemcpyAsync(h->h_pin + out_off[b], h->d_slab + (size_t)b * h->win_bytes + dio[b].out_off, dio[b].bytes - dio[b].out_off,
                               hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (h->prof) {
        slamit_ba_profile_out& O = h->prof_last;
        memset(&O, 0, sizeof(O));
        O.nwin = nwin; O.slots = (int32_t)(pev_used / 6);
        for (size_t s0 = 0; s0 + 6 <= pev_used; s0 += 6)
            for (int ph = 0; ph < SLAMIT_BA_PHASES; ++ph) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, h->pev[s0 + ph], h->pev[s0 + ph + 1]) == hipSuccess) O.phase_ms[ph] += ms;
            }
        for (int b = 0; b < nwin; ++b) O.schur_exec_mflop += exec_mflop[b];
    }
    auto unpack = [&](int b) {
        const slamit_ba_problem& P = probs[b];
        slamit_ba_result& R = results[b];
        // the output section as the host sees it: same carve, shifted so that its output part starts at out_off[b]
        const IoLayout H = carve_io(h->h_pin + out_off[b] - dio[b].out_off, P.n_kf, P.n_pt, P.n_edge, P.edge_ur != nullptr);
        memcpy(R.kf_pose, H.out_pose, sizeof(double) * 12 * (size_t)P.n_kf);
        for (int p2 = 0; p2 < P.n_pt; ++p2)   // points back in the caller's order
            for (int j = 0; j < 3; ++j) R.pt_xyz[3 * (size_t)perm[b][p2] + j] = H.out_pt[3 * (size_t)p2 + j];
        if (P.n_edge) {
            if (R.edge_chi2) memcpy(R.edge_chi2, H.out_chi2, sizeof(double) * (size_t)P.n_edge);
            if (R.edge_outlier) memcpy(R.edge_outlier, H.out_flag, (size_t)P.n_edge);
            if (R.edge_stage1_outlier) memcpy(R.edge_stage1_outlier, H.out_out1, (size_t)P.n_edge);
        }
        const BaState& S0 = *H.out_state;
        if (b == 0 && getenv("SLAMIT_BA_DIAG_WAVES")) fprintf(stderr, "[ba diag] busy cycles of waves 0..7: %llu %llu %llu %llu %llu %llu %llu %llu\n", S0.dbg[0], S0.dbg[1], S0.dbg[2], S0.dbg[3], S0.dbg[4], S0.dbg[5], S0.dbg[6], S0.dbg[7]);
        if (b == 0 && getenv("SLAMIT_BA_DIAG")) {  // diagnostic builds only: in-kernel clock and phases of the last LDLt launch
            fprintf(stderr, "[ba diag] ldlt shader cycles %llu, realtime ticks (100 MHz) %llu -> %.0f MHz, %.1f us\n",
                    S0.dbg[2] - S0.dbg[0], S0.dbg[3] - S0.dbg[1],
                    100.0 * (double)(S0.dbg[2] - S0.dbg[0]) / (double)(S0.dbg[3] - S0.dbg[1] + 1), (double)(S0.dbg[3] - S0.dbg[1]) / 100.0);
            fprintf(stderr, "[ba diag] ldlt phase cycles: load %llu factor %llu rows %llu writeback / wave-0 busy %llu trailing / rhs-wave busy %llu backsub %llu, pivot-wave busy %llu\n",
                    S0.dbg[4] >> 32, S0.dbg[4] & 0xffffffffull, S0.dbg[5] >> 32, S0.dbg[5] & 0xffffffffull, S0.dbg[6] >> 32, S0.dbg[6] & 0xffffffffull, S0.dbg[7]);
        }
        slamit_ba_stats* S = R.stats;
        if (!S) return;
        memset(S, 0, sizeof(*S));
        for (int sg = 0; sg < 2; ++sg) {
            S->n_its[sg] = S0.n_its[sg];
            S->chi2_init[sg] = S0.chi2_init[sg];
            for (int i = 0; i < SLAMIT_BA_MAX_ITS; ++i) { S->chi2[sg][i] = S0.chi2[sg][i]; S->lambda[sg][i] = S0.lam[sg][i]; S->trials[sg][i] = S0.trials[sg][i]; }
        }
    };
    {   // the windows' results are unpacked (points back into the caller's order) by the host threads that prepared them
        const int nthreads = std::max(1, std::min(std::min(nwin, 16), (int)std::thread::hardware_concurrency()));
        std::atomic<int> next(0);
        auto work = [&]() { for (int b = next.fetch_add(1); b < nwin; b = next.fetch_add(1)) unpack(b); };
        std::vector<std::thread> pool;
        try {
            for (int t = 1; t < nthreads && nwin >= 4; ++t) pool.emplace_back(work);
        } catch (...) {}
        work();
        for (std::thread& t : pool) t.join();
    }
    if (timing) {
        const double t_out = tclk();
        fprintf(stderr, "[ba timing] %d windows: validate %.3f | prepare + pack %.3f | queue uploads %.3f | LM loop %.3f | download + unpack %.3f ms\n",
                nwin, t_val - t_in, t_prep - t_val, t_queue - t_prep, t_loop - t_queue, t_out - t_loop);
    }
    return SLAMIT_OK;
}

int slamit_ba_solve(slamit_ba* h, const slamit_ba_problem* prob, const slamit_ba_opts* opts, slamit_ba_result* res) {
    return slamit_ba_solve_batch(h, 1, prob, opts, res);
}

int slamit_ba_profile(slamit_ba* h, int on) {
    if (!h) return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_profile: null handle");
    h->prof = on != 0;
    return SLAMIT_OK;
}

int slamit_ba_profile_read(slamit_ba* h, slamit_ba_profile_out* out) {
    if (!h || !out) return slamit_fail(SLAMIT_ERR_ARG, "slamit_ba_profile_read: bad argument");
    *out = h->prof_last;
    return SLAMIT_OK;
}

}  // extern "C"
