// ba_types.h — host/device shared layout of one local-BA window in HBM.
#ifndef SLAMIT_BA_TYPES_H
#define SLAMIT_BA_TYPES_H

#include <stdint.h>

#define BA_MAX_ITS 32       // == SLAMIT_BA_MAX_ITS
#define BA_TILE 64          // Schur macro tile (rows/cols of S per workgroup)
#define BA_KC 32            // k-depth of one staged slab
#define BA_SPLITS 16        // split-K factor of the Schur GEMM
#define BA_MAX_TILES 10     // 64-row tiles of the reduced system (slamit_ba_create refuses larger handles)
#define BA_MAX_PANELS 20    // 32-column LDLt panels
#define BA_JAC_MONO 21      // doubles per edge record: A (2x3) B (2x6) wO r0 r1
#define BA_JAC_STEREO 31    // ... of a window with stereo observations: A (3x3) B (3x6) wO r0 r1 r2 (a monocular edge's third row is zero)
#define BA_SF_MAXG 256      // groups of k slabs of the floating-window Schur product (BaWin::sf_*)
#define BA_SF_ROWS 63       // matrix rows one window holds (row 64 of its B operand is the right-hand side's row)
#define BA_SOLVER_BAND 0      // narrow row envelope: block LDLt inside LDS (k_ldlt_band)
#define BA_SOLVER_BLOCKED 1   // any other structure: 32-column panels through L2 (k_ldlt_blocked)

// LM control block of one window (device resident; the host only reads it back between chunks
// of enqueued trial slots).  Mirrors the locals of OptimizationAlgorithmLevenberg::solve
// (g2o/core/optimization_algorithm_levenberg.cpp:61-164).
struct BaState {
    int32_t done;            // stage finished (iterations exhausted / Terminate / nothing active)
    int32_t need_linearize;  // next slot starts a new iteration: Jacobians + normal equations
    int32_t it;              // iteration index inside the stage
    int32_t max_it;
    int32_t qmax;            // LM trials used so far in this iteration
    int32_t nBad;
    int32_t robust;          // Huber kernel on (stage 1) / off (stage 2)
    int32_t ok2;             // last factorisation succeeded
    int32_t n_active;        // active edges of the stage
    int32_t stage;
    double lambda, ni;
    double currentChi, iniChi, tempChi;
    unsigned long long maxdiag_bits;  // max |H_jj| as ordered bits (non-negative doubles)
    // per-stage statistics (slamit_ba_stats)
    int32_t n_its[2];
    double chi2[2][BA_MAX_ITS];
    double lam[2][BA_MAX_ITS];
    int32_t trials[2][BA_MAX_ITS];
    double chi2_init[2];
    unsigned long long dbg[8];  // diagnostic stamps (s_memtime / s_memrealtime); written only when BA_DIAG_STAMPS is defined
};

// One window.  All pointers are device addresses inside the handle's slabs.  In device code they carry the global
// address space: a plain pointer read out of a struct in memory is GENERIC to the compiler, and every access through it
// a flat_load (slower to issue, and it counts in lgkmcnt, so LDS waits also wait for outstanding HBM loads).
// (Only the kernel translation unit asks for it: host code that fills the struct is also parsed in the device pass.)
#if defined(BA_GLOBAL_POINTERS) && defined(__HIP_DEVICE_COMPILE__)
#define BA_G __attribute__((address_space(1)))
#else
#define BA_G

#endif
struct BaWin {
    int32_t n_kf, n_pt, n_edge, n_free;
    int32_t nS;        // 6 * n_free
    int32_t Npad;      // rows/cols of the padded reduced system, multiple of BA_TILE, >= nS + 1
    int32_t Kpad;      // padded k extent (3 * n_pt rounded up to BA_KC * BA_SPLITS)
    int32_t n_part;    // partial-sum slots of the chi2 reduction
    double huber_delta, chi2_gate;
    double huber_delta_s, chi2_gate_s;   // stereo edges (Optimizer.cc:570, :696)
    int32_t nrow;      // residual rows per edge record: 2 (every edge monocular) or 3 (the window has stereo observations)
    int32_t pad0;
    // Structure of the window (host, ba_api.hip), conservative for both stages.  Points are stored sorted by the first
    // free keyframe that observes them, so the non-zeros of a 64-row tile of the Schur operand GA sit in ONE k range, and the reduced
    // system has a row envelope (first coupled column per pose) that LDLt without pivoting never leaves.
    int32_t tile_alo[BA_MAX_TILES], tile_ahi[BA_MAX_TILES];   // k range (multiples of BA_KC) of GA's rows 64 t .. 64 t + 63
    int32_t tile_blo[BA_MAX_TILES], tile_bhi[BA_MAX_TILES];   // ... as the product's B operand (the tile that holds row nS, the right-hand side's row, spans every point)
    int16_t panel_hi[BA_MAX_PANELS];   // last matrix row with an entry in the 32 columns of LDLt panel i (>= the panel's last row)
    int16_t back_lo[BA_MAX_PANELS];    // first column any row of panel i's 32 rows reaches (back-substitution)
    int32_t band;      // half bandwidth of the reduced system's row envelope: max over rows r of r - first column of r (nS - 1: full)
    int32_t solver;    // which reduced solve takes the window (host, from nS and band): BA_SOLVER_*; one launch per kind present in a batch
    // Schur product over FLOATING row windows (host, ba_api.hip; sf_groups == 0: 64 x 64 tile pairs over k ranges as above).  The points are sorted
    // by their first observing keyframe, so the rows with non-zeros in one k slab of 32 (about eleven points) are a short run -- 48 rows
    // when every point is seen by eight consecutive keyframes.  Consecutive slabs whose rows fit ONE run of BA_SF_ROWS rows form a group:
    // one workgroup multiplies GA(rows, slabs) GA(rows + the right-hand side's row, slabs)^T into a 64 x 64 partial tile, and
    // k_schur_reduce adds, for an entry (r, c), the tiles of the groups whose window holds both rows -- a contiguous run of groups,
    // in group order.  Aligned 64 x 64 tile pairs multiply 3.8 x as many zeros on such a window.
    int32_t sf_groups, pad1;
    int16_t sf_row[BA_SF_MAXG];                       // first matrix row of group g's window
    int16_t sf_k0[BA_SF_MAXG], sf_k1[BA_SF_MAXG];     // its slabs [k0, k1), in units of BA_KC
    int16_t sf_glo[BA_MAX_TILES * BA_TILE], sf_ghi[BA_MAX_TILES * BA_TILE];   // per matrix row: the groups whose window holds it (glo > ghi: none)
    // vertices
    BA_G double* pose;      // n_kf x 7: q(x,y,z,w), t
    BA_G double* pose_bak;
    BA_G double* intr;      // n_kf x 4
    BA_G int32_t* pose_col; // n_kf: column block among free poses or -1
    BA_G double* pt;        // n_pt x 3
    BA_G double* pt_bak;
    // edges (caller order) + CSR by point and by free pose
    BA_G int32_t* e_kf; BA_G int32_t* e_pt;
    BA_G double* e_uv;      // n_edge x 2
    BA_G double* e_w;       // n_edge
    BA_G double* e_ur;      // n_edge: right-image column, < 0 on a monocular edge (stereo windows only)
    BA_G double* bf;        // n_kf: baseline x fx (stereo windows only)
    BA_G uint8_t* e_active;
    BA_G uint8_t* e_out1;   // stage-1 outlier flag
    BA_G double* e_chi2;    // chi2 of the last evaluated trial
    BA_G double* e_jac;     // n_edge x 21: A(2x3) B(2x6) wO r0 r1  (x 31 in a stereo window: BA_JAC_STEREO)
    BA_G int32_t* pt_ptr; BA_G int32_t* pt_edges;     // CSR: edges of each point
    BA_G int32_t* kf_ptr; BA_G int32_t* kf_edges;     // CSR: edges of each keyframe
    // normal equations
    BA_G double* Hll;       // n_pt x 6 (xx xy xz yy yz zz)
    BA_G double* bl;        // n_pt x 3
    BA_G double* Dinv;      // n_pt x 6
    BA_G double* Hpp;       // n_free x 36
    BA_G double* bp;        // n_free x 6
    BA_G double* GA;        // Npad x Kpad : the Schur operand G = Hpl Ci^T scattered (row = pose dof, col = 3*pt + j; Ci = inverse Cholesky factor of Hll + lambda I,
                            // ba_kernels.hip point_chol): S = Hpp - G G^T; row nS holds Ci bl, so that G (row nS)^T is the right-hand side's coefficient
    BA_G double* part;      // BA_SPLITS x Npad x Npad partial products
    BA_G double* S;         // Npad x Npad reduced system (symmetric, full)
    BA_G double* Sb;        // (Npad + 1) x 64: the same system as the banded solve's LDS image (rows of ldlt_band_rs(band) doubles; k_schur_reduce
                            // writes the in-band entries, the zeros around them are laid once per solve by k_import)
    BA_G double* rhs;       // Npad : b_schur in, x_pose out
    BA_G double* x_l;       // n_pt x 3 landmark increments
    BA_G double* chi_part;  // n_part partial robust-cost sums
    BA_G double* scale_part;// n_part partial sums of x(lambda x + b) over landmarks
    BA_G BaState* st;
};

// Where one window's inputs and outputs sit in its slab (device addresses): the host fills / reads them with ONE copy each.
struct BaIo {
    const double* in_pose;   // n_kf x 12 (R|t)
    const double* in_pt;     // n_pt x 3, device point order
    double* out_pose;        // n_kf x 12
    double* out_pt;          // n_pt x 3, device point order
    double* out_chi2;        // n_edge
    uint8_t* out_flag;       // n_edge
    uint8_t* out_out1;       // n_edge
    BaState* out_state;
};


#endif
