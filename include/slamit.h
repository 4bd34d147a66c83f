/*
 * slamit.h — C-ABI of libslamit_hip.so: the MI355X (gfx950) implementation of the ORB-SLAM2
 * per-frame hot path of serviceberry3/weiner_slamit_v2.
 *
 * Every entry point replaces one reference interface (paths relative to
 * oRB_SLAM2_Android/src/main/jni/ORB_SLAM2/ unless they start with Thirdparty/):
 *
 *   slamit_orb_*        <- ORBextractor::ORBextractor / operator()      include/ORBextractor.h:45-85,
 *                                                                        src/ORBextractor.cc:415-482,1064-1136
 *   slamit_orb_level    <- public member ORBextractor::mvImagePyramid    include/ORBextractor.h:85
 *   slamit_hamming_*    <- ORBmatcher::DescriptorDistance + best/second  src/ORBmatcher.cc:1651-1667, 85-117,
 *                          selection loops                               440-461, 1404-1428
 *   slamit_ba_*         <- Optimizer::LocalBundleAdjustment (the g2o     src/Optimizer.cc:453-778 and
 *                          BlockSolver_6_3 + Levenberg it instantiates)  Thirdparty/g2o/g2o/core/block_solver.hpp
 *
 * Conventions: plain C types only; `int` status return (0 = SLAMIT_OK, <0 = error, text from
 * slamit_last_error()); nothing throws across the boundary; the caller owns every buffer it
 * passes; the library owns device memory inside opaque handles.  Functions with the suffix
 * `_dev` take DEVICE pointers (HBM-resident buffers, e.g. a torch tensor's data_ptr) and a HIP
 * stream handle (`void*` = hipStream_t, NULL = the handle's own stream) and do not synchronise;
 * all others take HOST pointers and return when the result is in the caller's memory.
 * A handle may be used by one thread at a time; distinct handles are independent (the reference
 * runs two extractors on two threads for stereo, src/Frame.cc:93-94).
 */
#ifndef SLAMIT_H
#define SLAMIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SLAMIT_OK 0
#define SLAMIT_ERR_ARG (-1)      /* bad argument / unsupported geometry */
#define SLAMIT_ERR_DEVICE (-2)   /* HIP runtime error (no GPU, launch failure, out of memory) */
#define SLAMIT_ERR_CAPACITY (-3) /* caller buffer too small */
#define SLAMIT_ERR_STATE (-4)    /* call order (e.g. level requested before any extract) */

#define SLAMIT_DESC_BYTES 32
#define SLAMIT_EDGE_THRESHOLD 19 /* src/ORBextractor.cc:79 */

/* Field order and size identical to cv::KeyPoint (28 bytes) so the ORBextractor shim can
 * memcpy into std::vector<cv::KeyPoint>. */
typedef struct slamit_kp {
    float x, y;     /* level-0 pixel coordinates (already multiplied by the level scale) */
    float size;     /* (float)(int)(31 * scale[octave])        src/ORBextractor.cc:857,866 */
    float angle;    /* degrees in [0,360], fastAtan2 of IC     src/ORBextractor.cc:108 */
    float response; /* FAST-9/16 corner score */
    int32_t octave;
    int32_t class_id; /* always -1 */
} slamit_kp;

/* ---- ORB extractor --------------------------------------------------------------------- */

typedef struct slamit_orb_params {
    int32_t nfeatures;  /* 1000 (tracking) / 2000 (initialiser)  src/Tracking.cc:149,162 */
    float scale_factor; /* 1.2f */
    int32_t nlevels;    /* 8, at most SLAMIT_MAX_LEVELS */
    int32_t ini_th_fast; /* 20 */
    int32_t min_th_fast; /* 7 */
    int32_t width, height; /* frame geometry is fixed per handle */
    int32_t max_batch;     /* frames per extract_batch call the handle is sized for (>=1) */
} slamit_orb_params;

#define SLAMIT_MAX_LEVELS 16

typedef struct slamit_orb slamit_orb;

int slamit_orb_create(const slamit_orb_params* params, int device, slamit_orb** out);
void slamit_orb_destroy(slamit_orb* h);

/* Scale tables (GetScaleFactors & friends, include/ORBextractor.h:64-82). Each out array has
 * nlevels entries; any may be NULL. */
int slamit_orb_tables(const slamit_orb* h, float* scale, float* inv_scale, float* sigma2,
                      float* inv_sigma2, int32_t* features_per_level);

/* Upper bound on keypoints one frame can return: nfeatures + 3*nlevels (the octree may
 * overshoot each level's quota by up to 3, src/ORBextractor.cc:743-744). */
int slamit_orb_max_keypoints(const slamit_orb* h);

/* One frame, host buffers: gray is h rows of `stride` bytes. Writes *n_out keypoints
 * (level-major order as the reference concatenates them) and n_out*32 descriptor bytes.
 * cap must be >= slamit_orb_max_keypoints(). Empty image (w or h == 0 at create) -> n_out = 0. */
int slamit_orb_extract(slamit_orb* h, const uint8_t* gray, size_t stride, slamit_kp* kps,
                       uint8_t* desc, int cap, int* n_out);

/* nframes frames, host buffers; frame f at gray + f*frame_stride; outputs for frame f at
 * kps + f*cap, desc + f*cap*32, n_out[f]. */
int slamit_orb_extract_batch(slamit_orb* h, const uint8_t* gray, size_t stride,
                             size_t frame_stride, int nframes, slamit_kp* kps, uint8_t* desc,
                             int cap, int* n_out);

/* Same, device buffers (all pointers are HBM addresses), asynchronous on `stream`. */
int slamit_orb_extract_batch_dev(slamit_orb* h, const uint8_t* d_gray, size_t stride,
                                 size_t frame_stride, int nframes, slamit_kp* d_kps,
                                 uint8_t* d_desc, int cap, int32_t* d_n_out, void* stream);

/* Per-stage device timing with HIP events recorded on the launch stream (the counterpart of
 * g2o's G2OBatchStatistics idea, Thirdparty/g2o/g2o/core/batch_stats.h:39-77, for the extractor).
 * Reads (and clears) what was accumulated since the last call into stage_ms / stage_calls
 * (either may be NULL), then switches recording on/off.  Stages: 0 pyramid resize, 1 FAST cells,
 * 2 octree, 3 IC angle, 4 Gaussian blur, 5 rBRIEF + output.  A "call" is one stage of one
 * extract_batch call (the resize stage launches one kernel per level).  enable: 0 off, 1 every stage,
 * n >= 2 only stage 1 (the dominant kernel) on every (n-1)-th extract call: an event between two kernels costs
 * ~10 us of pipeline drain, so a throughput run that still wants the dominant kernel's live duration samples it. */
#define SLAMIT_ORB_STAGES 6
int slamit_orb_profile(slamit_orb* h, int enable, float* stage_ms, int32_t* stage_calls, int nstages);

/* mvImagePyramid[level] of frame `frame` of the last extract call: copies the padded plane
 * ((w+38) x (h+38), REFLECT_101 border of 19) to host memory. dst may be NULL to query sizes;
 * *w,*h are the un-padded level size, the plane is (*h+38) rows of (*w+38) bytes. */
int slamit_orb_level(slamit_orb* h, int frame, int level, uint8_t* dst, size_t dst_bytes, int* w,
                     int* h_out);

/* Stage outputs of the last extract call, for parity debugging (host buffers):
 *  candidates of (frame, level) before the octree as (x, y, score) int32 triplets, x/y relative
 *  to the (16,16) detection border like vToDistributeKeys (src/ORBextractor.cc:840-845), sorted
 *  in the reference's (cell row, cell col, y, x) order. */
int slamit_orb_debug_candidates(slamit_orb* h, int frame, int level, int32_t* xys, int cap,
                                int* n_out);
/*  the blurred level the descriptors were sampled from: GaussianBlur(mvImagePyramid[level].clone(), 7x7, sigma 2,
 *  BORDER_REFLECT_101) of src/ORBextractor.cc:1116-1117, *h_out rows of *w bytes (no border). dst may be NULL
 *  to query the size. */
int slamit_orb_debug_blurred(slamit_orb* h, int frame, int level, uint8_t* dst, size_t dst_bytes, int* w,
                             int* h_out);

/* ---- Hamming matcher -------------------------------------------------------------------- */

#define SLAMIT_HAMMING_MAX_TRAIN 65535   /* train rows per set in the best/second entry points (index packed in 16 bits) */
/* For each of nq query descriptors: best and second-best Hamming distance over the train
 * descriptors, and the index of the best (strict '<', first index wins; the reference's
 * selection rule, src/ORBmatcher.cc:1404-1428). With nt == 0: best = second = 256, idx = -1. */
int slamit_hamming_best2(const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* best_idx,
                         int32_t* best, int32_t* second);

/* Batched device form: pair p matches d_q + p*q_stride (nq[p] rows) against d_t + p*t_stride
 * (nt[p] rows); outputs at p*out_stride. d_nq / d_nt are device int32 arrays (they are the
 * d_n_out of extract_batch_dev), max_n bounds both. */
int slamit_hamming_best2_batch_dev(const uint8_t* d_q, const int32_t* d_nq, size_t q_stride,
                                   const uint8_t* d_t, const int32_t* d_nt, size_t t_stride,
                                   int npairs, int max_n, int32_t* d_best_idx, int32_t* d_best,
                                   int32_t* d_second, size_t out_stride, int device, void* stream);

/* MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:248-314), batched:
 * map point p owns descriptor rows [offsets[p], offsets[p+1]) of `desc` (its observations, in the
 * reference's iteration order).  For each point: all-pairs Hamming distances, per row the median
 * vDists[(int)(0.5*(N-1))] of the sorted row (self distance 0 included), and the row with the least
 * median (first one on ties) -> best_idx[p] (relative to offsets[p]; -1 for a point without rows),
 * best_median[p].  At most SLAMIT_DISTINCTIVE_MAX rows per point. */
#define SLAMIT_DISTINCTIVE_MAX 128
int slamit_distinctive_batch(const uint8_t* desc, const int32_t* offsets, int npoints, int32_t* best_idx,
                             int32_t* best_median);

/* ---- Guided search ------------------------------------------------------
 * The common core of ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th)
 * (src/ORBmatcher.cc:47-131) and SearchByProjection(CurrentFrame, LastFrame, th, bMono) (:1332-1474):
 * for each query, in order, Frame::GetFeaturesInArea(u, v, r, minLevel, maxLevel)
 * (src/Frame.cc:447-502 over the 64x48 grid of Frame::AssignFeaturesToGrid, :336-357, :505-517),
 * skip keypoints that already carry a map point, take the nearest (and second nearest) descriptor in
 * the reference's candidate order, accept, and MARK THE KEYPOINT TAKEN for the queries that follow
 * (the reference assigns F.mvpMapPoints[bestIdx] inside the loop).  Projection, viewing-cosine radius
 * and the rotation histogram stay with the caller (shim/ORBmatcher.h).  Mono only (mvuRight < 0). */
typedef struct slamit_frame_view {
    int32_t n;                 /* keypoints */
    const float* kp_xy;        /* n x 2: mvKeysUn[i].pt */
    const int32_t* kp_octave;  /* n */
    const uint8_t* desc;       /* n x 32: mDescriptors */
    const uint8_t* kp_taken;   /* n: 1 = mvpMapPoints[i] && mvpMapPoints[i]->Observations() > 0 on entry */
    float min_x, min_y;        /* mnMinX, mnMinY */
    float inv_w, inv_h;        /* mfGridElementWidthInv, mfGridElementHeightInv */
} slamit_frame_view;

typedef struct slamit_search_queries {
    int32_t m;
    const float* uvr;          /* m x 3: window centre u, v and half-size r (already scaled) */
    const int32_t* level_min;  /* m */
    const int32_t* level_max;  /* m: -1 = no upper bound (GetFeaturesInArea's default) */
    const uint8_t* desc;       /* m x 32: pMP->GetDescriptor() */
    const uint8_t* valid;      /* m: 0 = query skipped (mbTrackInView false, bad point, behind camera ...) */
    const uint8_t* takes;      /* m or NULL (= all 1): 1 = pMP->Observations() > 0, i.e. a keypoint matched to
                                  this query is skipped by the queries that follow */
} slamit_search_queries;

typedef struct slamit_search_rule {
    int32_t th_dist;           /* accept iff bestDist <= th_dist (TH_HIGH = 100, TH_LOW = 50) */
    int32_t use_ratio;         /* 1: also reject when bestLevel == bestLevel2 && bestDist > nnratio * bestDist2 */
    float nnratio;
    /* ORBmatcher::Fuse's per-candidate gate (src/ORBmatcher.cc:925-936): with e2 = (u - kp.x)^2 + (v - kp.y)^2 a
     * candidate is skipped when e2 * inv_level_sigma2[kp.octave] > chi2_gate (5.99).  chi2_gate <= 0: no gate. */
    float chi2_gate;
    float inv_level_sigma2[16];
    /* mode 0: the SearchByProjection / Fuse loop described above.
     * mode 1: ORBmatcher::SearchForInitialization (src/ORBmatcher.cc:409-474): instead of the taken flag a keypoint
     *   carries the distance of its current match; a candidate is skipped when that distance is <= the query's
     *   distance to it (:448); accept iff bestDist <= th_dist && bestDist < (float)bestDist2 * nnratio (bestDist2 =
     *   INT_MAX without a second candidate); an accepted query takes the keypoint over from the query that held it,
     *   whose match_kp entry goes back to -1.  kp_taken / takes / use_ratio / chi2_gate are ignored.  best_level then
 *   reports the keypoint a query was matched to AT ITS OWN TURN (-1 if it was not accepted), which a later take-over
 *   does not reset: the reference bins exactly those into its rotation histogram (:467-477). */
    int32_t mode;
} slamit_search_rule;

/* match_kp[q] = index of the keypoint the query took, or -1; *nmatches = number of accepted queries.
 * best_dist / best_level / second_dist / second_level (any may be NULL) report the selection of every
 * query that had candidates (256 / -1 otherwise).  At most SLAMIT_SEARCH_MAX_KP keypoints per frame.  A window may hold any
 * number of them: SLAMIT_SEARCH_MAX_CAND is only the length of the stored candidate list a query's re-scan reads; a query
 * with more candidates whose tentative pair was taken by an earlier query walks the frame's keypoints again. */
#define SLAMIT_SEARCH_MAX_KP 8191
#define SLAMIT_SEARCH_MAX_CAND 1024
int slamit_guided_search(int device, const slamit_frame_view* frame, const slamit_search_queries* queries,
                         const slamit_search_rule* rule, int32_t* match_kp, int32_t* nmatches, int32_t* best_dist,
                         int32_t* best_level, int32_t* second_dist, int32_t* second_level);

/* ---- Vocabulary-node search (beyond SURVEY.md §8f: the BoW drivers of ORBmatcher) ----------------------
 * The loop bodies of ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) (src/ORBmatcher.cc:161-290),
 * ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...) (:526-657) and ORBmatcher::SearchForTriangulation (:659-826):
 * features of the two sides that fall into the same vocabulary node (DBoW2::FeatureVector, built by the reference's
 * vendored DBoW2 on the host) are compared with DescriptorDistance.  A group = one node present on both sides:
 * its side-1 feature indices in processing order (queries) and its side-2 feature indices in scan order (candidates).
 * A feature belongs to one node, so groups are independent (checked: an index may appear once per side); one
 * wavefront walks a group's queries in order.
 *
 * mode 0 (SearchByBoW): per query best / second best with strict '<' over the candidates that are allowed (valid2)
 *   and not yet matched by an earlier query; accepted iff best <= th (th_inclusive, :243) or best < th (:601) and
 *   (float)best < nnratio * (float)second (second = 256 without one); an accepted query takes its candidate.
 * mode 1 (SearchForTriangulation, monocular): per query the candidate of minimum distance among those with
 *   dist <= th that pass the epipole test (:737-743) and CheckDistEpipolarLine (:135-158), the LAST such candidate on
 *   ties (:731 'dist > bestDist' lets an equal one replace); candidates are never marked (the reference declares
 *   vbMatched2 but does not set it).  Float expressions are evaluated as written, without contraction.
 * The rotation-histogram filter that follows in all three drivers (mbCheckOrientation) is host logic (shim). */
typedef struct slamit_bow_groups {
    int32_t n_groups;
    const int32_t* q_ptr;      /* n_groups + 1: queries of group g are q_idx[q_ptr[g] .. q_ptr[g+1]) */
    const int32_t* q_idx;      /* side-1 feature indices */
    const int32_t* c_ptr;      /* n_groups + 1 */
    const int32_t* c_idx;      /* side-2 feature indices */
} slamit_bow_groups;

typedef struct slamit_bow_rule {
    int32_t mode;              /* 0 SearchByBoW, 1 SearchForTriangulation */
    int32_t th;                /* TH_LOW = 50 */
    int32_t th_inclusive;      /* mode 0: 1 = best <= th (KeyFrame/Frame), 0 = best < th (KeyFrame/KeyFrame) */
    float nnratio;             /* mode 0 */
    /* mode 1 only */
    float F12[9];              /* fundamental matrix, row-major */
    float ex, ey;              /* epipole of camera 1 in image 2 */
    const float* kp1_xy;       /* n1 x 2: mvKeysUn of side 1 */
    const float* kp2_xy;       /* n2 x 2 */
    const int32_t* kp2_octave; /* n2 */
    float scale_factor[16];    /* pKF2->mvScaleFactors */
    float level_sigma2[16];    /* pKF2->mvLevelSigma2 */
} slamit_bow_rule;

#define SLAMIT_BOW_MAX_GROUP 2048   /* candidates of one group */
/* valid1[i] != 0: side-1 feature i is a query (has a good MapPoint / has none yet); valid2[i] != 0: side-2 feature i
 * may be matched; either may be NULL (= all).  match12[i] = matched side-2 index or -1, dist12[i] (may be NULL) = the
 * best distance query i saw (256 if none); both have n1 entries.  *nmatches = accepted queries. */
int slamit_bow_search(int device, const uint8_t* desc1, int32_t n1, const uint8_t* valid1, const uint8_t* desc2, int32_t n2,
                      const uint8_t* valid2, const slamit_bow_groups* groups, const slamit_bow_rule* rule, int32_t* match12,
                      int32_t* dist12, int32_t* nmatches);

/* ---- Frame epilogue -------------------------------------------------------
 * What Frame's constructors do right after the extractor: Frame::UndistortKeyPoints (src/Frame.cc:529-559, through
 * cv::undistortPoints(mat, mat, mK, mDistCoef, cv::Mat(), mK)) and Frame::AssignFeaturesToGrid (:336-357, PosInGrid
 * :505-517), fused so that keypoints never leave the GPU between extraction and the guided search.
 * cv::undistortPoints is OpenCV's (absent from the reference tree): restated from the published cvUndistortPoints,
 * parity unpinned (oracle/orb_oracle.cc).  k1 == 0 leaves the keypoints untouched like Frame.cc:531-535. */
typedef struct slamit_camera {
    float fx, fy, cx, cy;      /* mK */
    float k1, k2, p1, p2, k3;  /* mDistCoef (k3 = 0 for a 4-coefficient model) */
} slamit_camera;

#define SLAMIT_FRAME_GRID_COLS 64   /* FRAME_GRID_COLS, include/Frame.h:41 */
#define SLAMIT_FRAME_GRID_ROWS 48   /* FRAME_GRID_ROWS, include/Frame.h:40 */
#define SLAMIT_FRAME_GRID_CELLS (SLAMIT_FRAME_GRID_COLS * SLAMIT_FRAME_GRID_ROWS)
#define SLAMIT_FRAME_MAX_KP 30000

/* cv::undistortPoints(xy, xy, K, D, Mat(), K) on n points (what Frame::ComputeImageBounds feeds the four image
 * corners to, Frame.cc:561-590).  No k1 == 0 shortcut here: that belongs to Frame::UndistortKeyPoints. */
int slamit_undistort_points(int device, const slamit_camera* cam, const float* xy_in, int n, float* xy_out);

/* kps_un[i] = kps[i] with the undistorted pt (mvKeysUn); the grid mGrid[x][y] as CSR: the indices of cell
 * c = x * 48 + y are cell_items[cell_start[c] .. cell_start[c+1]) in keypoint (push_back) order, cell_start has
 * SLAMIT_FRAME_GRID_CELLS + 1 entries, cell_start[last] = number of keypoints inside the grid.
 * min_x/min_y/inv_w/inv_h = mnMinX, mnMinY, mfGridElementWidthInv, mfGridElementHeightInv. */
int slamit_frame_finish(int device, const slamit_camera* cam, const slamit_kp* kps, int n, float min_x, float min_y,
                        float inv_w, float inv_h, slamit_kp* kps_un, int32_t* cell_start, int32_t* cell_items);

/* Same for a batch of frames resident in HBM in the layout slamit_orb_extract_batch_dev writes (frame f: d_kps +
 * f * cap, d_n[f] keypoints); d_cell_start is [nframes][SLAMIT_FRAME_GRID_CELLS + 1], d_cell_items [nframes][cap].
 * Asynchronous on `stream`. */
int slamit_frame_finish_batch_dev(int device, const slamit_camera* cam, const slamit_kp* d_kps, const int32_t* d_n, int cap,
                                  int nframes, float min_x, float min_y, float inv_w, float inv_h, slamit_kp* d_kps_un,
                                  int32_t* d_cell_start, int32_t* d_cell_items, void* stream);

/* The same for a batch of frames whose data is resident in HBM (one wavefront walks each frame's queries, all frames
 * in parallel): keypoints in the layout slamit_frame_finish_batch_dev writes, everything else [nframes][cap] strided.
 * SLAMIT_SEARCH_BATCH_CAND candidates are stored per query (the workspace's size); windows with more are still exact: a re-scan of
 * such a query walks the frame's keypoints again (the reference has no limit, ORBmatcher.cc:85-117). */
#define SLAMIT_SEARCH_BATCH_CAND 128
typedef struct slamit_search_batch {
    int32_t nframes, kp_cap, q_cap;
    const int32_t* d_n;            /* [nframes] keypoints per frame */
    const slamit_kp* d_kps_un;     /* [nframes][kp_cap] (pt and octave are read) */
    const uint8_t* d_desc;         /* [nframes][kp_cap][32] */
    const uint8_t* d_kp_taken;     /* [nframes][kp_cap] */
    float min_x, min_y, inv_w, inv_h;
    const int32_t* d_m;            /* [nframes] queries per frame */
    const float* d_uvr;            /* [nframes][q_cap][3] */
    const int32_t* d_level_min;    /* [nframes][q_cap] */
    const int32_t* d_level_max;
    const uint8_t* d_qdesc;        /* [nframes][q_cap][32] */
    const uint8_t* d_valid;        /* [nframes][q_cap] */
    const uint8_t* d_takes;        /* [nframes][q_cap] */
} slamit_search_batch;
size_t slamit_guided_search_workspace(int nframes, int q_cap);
/* d_match_kp [nframes][q_cap], d_nmatches [nframes], d_out4 [nframes][q_cap][4] or NULL (best dist / level, second
 * dist / level).  Asynchronous on `stream`. */
int slamit_guided_search_batch_dev(int device, const slamit_search_batch* batch, const slamit_search_rule* rule,
                                   int32_t* d_match_kp, int32_t* d_nmatches, int32_t* d_out4, void* d_workspace,
                                   size_t workspace_bytes, void* stream);

/* Full distance matrix (nq x nt, uint16), the batched form of DescriptorDistance. */
int slamit_hamming_matrix(const uint8_t* q, int nq, const uint8_t* t, int nt, uint16_t* out);

/* ---- Local bundle adjustment ------------------------------------------------------------ */

typedef struct slamit_ba_problem {
    int32_t n_kf;           /* local + fixed keyframes */
    int32_t n_pt;
    int32_t n_edge;
    const double* kf_pose;  /* n_kf x 12: R row-major (9) then t (3), world->camera, already widened
                               from float like Converter::toSE3Quat (src/Converter.cc:37-47) */
    const uint8_t* kf_fixed; /* n_kf: 1 = fixed vertex (KF id 0 or lFixedCameras) */
    const double* kf_intr;  /* n_kf x 4: fx fy cx cy */
    const double* pt_xyz;   /* n_pt x 3 */
    const int32_t* edge_kf; /* n_edge, in the reference's insertion order (per point, per observation) */
    const int32_t* edge_pt; /* n_edge */
    const double* edge_uv;  /* n_edge x 2 */
    const double* edge_inv_sigma2; /* n_edge */
    /* Stereo observations (EdgeStereoSE3ProjectXYZ, Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:112-141; built by
       src/Optimizer.cc:621-650).  Both NULL: every edge is monocular.  Otherwise edge_ur[e] is the keypoint's column in
       the right image (KeyFrame::mvuRight), negative for a monocular edge (the test at src/Optimizer.cc:596), and
       kf_bf[k] is keyframe k's baseline x fx (KeyFrame::mbf). */
    const double* edge_ur;  /* n_edge, nullable */
    const double* kf_bf;    /* n_kf, nullable (required when edge_ur is given) */
} slamit_ba_problem;

typedef struct slamit_ba_opts {
    int32_t its_robust;     /* 5   src/Optimizer.cc:660 */
    int32_t its_final;      /* 10  src/Optimizer.cc:707 */
    double huber_delta;     /* (double)(float)sqrt(5.991)  src/Optimizer.cc:569 */
    double chi2_gate;       /* 5.991 src/Optimizer.cc:680,723 */
    const volatile uint8_t* stop; /* nullable; polled like SparseOptimizer::terminate() */
    double huber_delta_stereo; /* (double)(float)sqrt(7.815)  src/Optimizer.cc:570; <= 0: that default */
    double chi2_gate_stereo;   /* 7.815 src/Optimizer.cc:696,740; <= 0: that default */
} slamit_ba_opts;

#define SLAMIT_BA_MAX_ITS 32

typedef struct slamit_ba_stats {
    int32_t n_its[2];                       /* LM iterations run in stage 1 / stage 2 */
    double chi2[2][SLAMIT_BA_MAX_ITS];      /* robust cost after each iteration */
    double lambda[2][SLAMIT_BA_MAX_ITS];    /* lambda after each iteration */
    int32_t trials[2][SLAMIT_BA_MAX_ITS];   /* LM trials used by each iteration */
    double chi2_init[2];                    /* cost before the first iteration of each stage */
} slamit_ba_stats;

typedef struct slamit_ba_result {
    double* kf_pose;        /* n_kf x 12 out */
    double* pt_xyz;         /* n_pt x 3 out */
    double* edge_chi2;      /* n_edge out: chi2 at the final estimate */
    uint8_t* edge_outlier;  /* n_edge out: 1 = chi2 > gate or depth <= 0 at the end (vToErase) */
    uint8_t* edge_stage1_outlier; /* n_edge out: 1 = removed after the robust stage (setLevel(1)) */
    slamit_ba_stats* stats; /* nullable */
} slamit_ba_result;

typedef struct slamit_ba slamit_ba;

/* A BA handle owns device workspaces sized for up to max_kf/max_pt/max_edge and max_batch
 * independent windows. */
int slamit_ba_create(int max_kf, int max_pt, int max_edge, int max_batch, int device,
                     slamit_ba** out);
void slamit_ba_destroy(slamit_ba* h);

/* One window, host buffers, synchronous. */
int slamit_ba_solve(slamit_ba* h, const slamit_ba_problem* prob, const slamit_ba_opts* opts,
                    slamit_ba_result* res);

/* nwin independent windows solved concurrently (one workgroup cluster per window). */
int slamit_ba_solve_batch(slamit_ba* h, int nwin, const slamit_ba_problem* probs,
                          const slamit_ba_opts* opts, slamit_ba_result* results);

/* Per-phase device time of the LM trial slots -- the analogue of g2o's G2OBatchStatistics (Thirdparty/g2o/g2o/core/batch_stats.h:39-77:
 * timeLinearize + timeQuadraticForm, timeSchurComplement, timeLinearSolver, timeUpdate, timeResiduals), which the reference never
 * switches on.  After slamit_ba_profile(h, 1) every solve on the handle records HIP events at the phase boundaries of each slot
 * (an event between two kernels drains the pipeline: such a solve runs slower and is not for timed runs);
 * slamit_ba_profile_read returns the sums over the slots of the LAST profiled solve (all windows of a batch run a phase in one launch). */
#define SLAMIT_BA_PHASES 5
typedef struct slamit_ba_profile_out {
    double phase_ms[SLAMIT_BA_PHASES]; /* 0 linearise + quadratic form + damping, 1 Schur complement (product + reduction),
                                          2 reduced solve (LDLt), 3 update (back-substitution + oplus), 4 residuals + LM decision */
    int32_t slots;                     /* LM trial slots queued */
    int32_t nwin;
    double schur_exec_mflop;           /* flops (1e6) the Schur product executes per trial, summed over the windows: its tile granules,
                                          against the algorithmic count of SURVEY.md 8(d) */
} slamit_ba_profile_out;
int slamit_ba_profile(slamit_ba* h, int on);
int slamit_ba_profile_read(slamit_ba* h, slamit_ba_profile_out* out);

/* ---- Pose-only optimisation (SURVEY.md §8f "next" rank 1) ----------------------------------
 * Optimizer::PoseOptimization (src/Optimizer.cc:239-451): one SE3 pose, n unary reprojection edges
 * (g2o EdgeSE3ProjectXYZOnlyPose, Thirdparty/g2o/g2o/types/types_six_dof_expmap.{h:143-170,cpp:266-288}),
 * four rounds of 10 Levenberg-Marquardt iterations, each restarted from the INPUT pose over the
 * current inliers; after every round an edge is an outlier iff (float)chi2 > 5.991f; the Huber
 * kernel is dropped after the third round.  Runs in one workgroup per frame, whole schedule on the
 * device.  Returns through n_inliers what the reference returns (nInitialCorrespondences - nBad;
 * 0 and an untouched pose when n < 3). */
typedef struct slamit_pose_problem {
    int32_t n;               /* correspondences (map point <-> undistorted keypoint) */
    const double* pose;      /* 12: R row-major, t — pFrame->mTcw widened like Converter::toSE3Quat */
    const double* intr;      /* 4: fx fy cx cy */
    const double* xw;        /* n x 3 world points (float positions widened) */
    const double* uv;        /* n x 2 */
    const double* inv_sigma2;/* n */
    /* Stereo correspondences (EdgeStereoSE3ProjectXYZOnlyPose, Thirdparty/g2o/g2o/types/types_six_dof_expmap.h:174-202;
       src/Optimizer.cc:319-356): ur[i] = the keypoint's column in the right image (Frame::mvuRight), negative for a
       monocular one; bf = Frame::mbf.  ur NULL: every correspondence is monocular and bf is not read. */
    const double* ur;        /* n, nullable */
    double bf;
} slamit_pose_problem;

typedef struct slamit_pose_result {
    double* pose;            /* 12 out */
    uint8_t* outlier;        /* n out: pFrame->mvbOutlier after the last round */
    int32_t n_inliers;       /* out */
    int32_t n_its[4];        /* out: LM iterations run in each round */
    double chi2[4];          /* out: robust cost of the last evaluated trial of each round */
} slamit_pose_result;

/* nframes independent frames in one launch (host pointers, synchronous). */
int slamit_pose_optimize_batch(int device, int nframes, const slamit_pose_problem* probs, slamit_pose_result* results);
int slamit_pose_optimize(int device, const slamit_pose_problem* prob, slamit_pose_result* res);

/* ---- Sim3 between two keyframes (beyond SURVEY.md §8f: loop closing) ---------------------------------
 * Optimizer::OptimizeSim3 (src/Optimizer.cc:1046-1247) with the g2o it instantiates: one VertexSim3Expmap
 * (Thirdparty/g2o/g2o/types/types_seven_dof_expmap.h, sim3.h), per correspondence a fixed point in each camera frame and
 * the pair EdgeSim3ProjectXYZ (x1 = K1 proj(S12 X2)) / EdgeInverseSim3ProjectXYZ (x2 = K2 proj(S12^-1 X1)), Huber kernels
 * of width (float)sqrt(th2), NUMERIC Jacobians (g2o's central differences, delta 1e-9, core/base_binary_edge.hpp:131-200:
 * the analytic ones are commented out in the reference), Levenberg-Marquardt on the dense 7 x 7 system.  Schedule:
 * 5 iterations, drop every pair with chi2 > th2 on either edge, return 0 if fewer than 10 pairs are left, 10 more
 * iterations (5 if nothing was dropped), count the pairs with both chi2 <= th2.  One workgroup per problem. */
typedef struct slamit_sim3_problem {
    int32_t n;                   /* correspondences that pass the reference's validity tests (:1112-1136) */
    const double* p1;            /* n x 3: P3D1c = R1w P1 + t1w (float values widened) */
    const double* p2;            /* n x 3: P3D2c */
    const double* obs1;          /* n x 2: kpUn1.pt */
    const double* obs2;          /* n x 2: kpUn2.pt */
    const double* inv_sigma2_1;  /* n: pKF1->mvInvLevelSigma2[kpUn1.octave] */
    const double* inv_sigma2_2;  /* n */
    double intr1[4], intr2[4];   /* fx fy cx cy of K1, K2 */
    double r12[9], t12[3], s12;  /* g2oS12 on entry: rotation (row-major), translation, scale */
    double th2;                  /* chi2 threshold (10 in LoopClosing::ComputeSim3) */
    int32_t fix_scale;           /* bFixScale */
} slamit_sim3_problem;

typedef struct slamit_sim3_result {
    double r12[9], t12[3], s12;  /* optimised g2oS12 (the input when the function returns 0 at the 10-pair test) */
    uint8_t* inlier;             /* n out: 0 = the reference sets vpMatches1[idx] to NULL */
    int32_t n_inliers;           /* the reference's return value */
    int32_t n_its[2];            /* LM iterations run in each stage */
    double chi2[2];              /* robust cost of the last evaluated trial of each stage */
} slamit_sim3_result;

int slamit_sim3_optimize_batch(int device, int nproblems, const slamit_sim3_problem* probs, slamit_sim3_result* results);
int slamit_sim3_optimize(int device, const slamit_sim3_problem* prob, slamit_sim3_result* res);

/* ---- misc -------------------------------------------------------------------------------- */

const char* slamit_last_error(void);
const char* slamit_version(void);
int slamit_device_count(void);

/* Devices and streams.  Every entry point runs on the device it is given (handles: the device of slamit_*_create) and
 * restores the caller's current device before it returns.  The host-pointer entry points WITHOUT a device argument
 * (slamit_hamming_best2, slamit_hamming_matrix, slamit_distinctive_batch) use the calling thread's slamit_set_device()
 * (-1 = whatever device is current, the default).  `stream` arguments: NULL means the handle's own stream for entry points
 * that take a handle, and the legacy default stream of `device` for the handle-less *_dev entry points; pass an explicit
 * stream to order several calls.  Host-pointer entry points keep one pinned block, one device slab and one stream per
 * calling thread and call site; they are released when the thread exits or by slamit_release_thread_scratch(). */
int slamit_set_device(int device);
void slamit_release_thread_scratch(void);

#ifdef __cplusplus
}
#endif
#endif /* SLAMIT_H */
