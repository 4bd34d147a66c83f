"""Python binding of libslamit_hip.so (the C-ABI of include/slamit.h) — used by tests and bench.py.

The classes keep the reference's names and argument meaning:
  ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)   include/ORBextractor.h:50-57
  ORBmatcher.DescriptorDistance / best2 / TH_LOW / TH_HIGH              include/ORBmatcher.h:41-89
  Optimizer.LocalBundleAdjustment                                        include/Optimizer.h:45
There is NO CPU fallback: if the library is missing or no GPU is usable every call raises.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLAMIT_LIB", os.path.join(HERE, "libslamit_hip.so"))  # override only for diagnostic builds

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")]
)

MAX_ITS = 32


class SlamitError(RuntimeError):
    pass


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("max_batch", C.c_int32)]


class BaProblem(C.Structure):
    _fields_ = [("n_kf", C.c_int32), ("n_pt", C.c_int32), ("n_edge", C.c_int32),
                ("kf_pose", C.c_void_p), ("kf_fixed", C.c_void_p), ("kf_intr", C.c_void_p),
                ("pt_xyz", C.c_void_p), ("edge_kf", C.c_void_p), ("edge_pt", C.c_void_p),
                ("edge_uv", C.c_void_p), ("edge_inv_sigma2", C.c_void_p),
                ("edge_ur", C.c_void_p), ("kf_bf", C.c_void_p)]   # stereo observations: both NULL for a monocular window


class BaOpts(C.Structure):
    _fields_ = [("its_robust", C.c_int32), ("its_final", C.c_int32), ("huber_delta", C.c_double),
                ("chi2_gate", C.c_double), ("stop", C.c_void_p),
                ("huber_delta_stereo", C.c_double), ("chi2_gate_stereo", C.c_double)]   # 0: the reference's sqrt(7.815) / 7.815


class BaStats(C.Structure):
    _fields_ = [("n_its", C.c_int32 * 2), ("chi2", (C.c_double * MAX_ITS) * 2),
                ("lambda_", (C.c_double * MAX_ITS) * 2), ("trials", (C.c_int32 * MAX_ITS) * 2),
                ("chi2_init", C.c_double * 2)]


class BaResult(C.Structure):
    _fields_ = [("kf_pose", C.c_void_p), ("pt_xyz", C.c_void_p), ("edge_chi2", C.c_void_p),
                ("edge_outlier", C.c_void_p), ("edge_stage1_outlier", C.c_void_p), ("stats", C.c_void_p)]


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("kp_xy", C.c_void_p), ("kp_octave", C.c_void_p), ("desc", C.c_void_p),
                ("kp_taken", C.c_void_p), ("min_x", C.c_float), ("min_y", C.c_float), ("inv_w", C.c_float),
                ("inv_h", C.c_float)]


class SearchQueries(C.Structure):
    _fields_ = [("m", C.c_int32), ("uvr", C.c_void_p), ("level_min", C.c_void_p), ("level_max", C.c_void_p),
                ("desc", C.c_void_p), ("valid", C.c_void_p), ("takes", C.c_void_p)]


class SearchRule(C.Structure):
    _fields_ = [("th_dist", C.c_int32), ("use_ratio", C.c_int32), ("nnratio", C.c_float), ("chi2_gate", C.c_float),
                ("inv_level_sigma2", C.c_float * 16), ("mode", C.c_int32)]


def _search_rule(th_dist, use_ratio, nnratio, chi2_gate=0.0, inv_level_sigma2=None, mode=0):
    sig = [1.0] * 16
    if inv_level_sigma2 is not None:
        for i, v in enumerate(list(inv_level_sigma2)[:16]):
            sig[i] = float(v)
    return SearchRule(int(th_dist), int(bool(use_ratio)), float(nnratio), float(chi2_gate), (C.c_float * 16)(*sig), int(mode))


class BowGroups(C.Structure):
    _fields_ = [("n_groups", C.c_int32), ("q_ptr", C.c_void_p), ("q_idx", C.c_void_p), ("c_ptr", C.c_void_p), ("c_idx", C.c_void_p)]


class BowRule(C.Structure):
    _fields_ = [("mode", C.c_int32), ("th", C.c_int32), ("th_inclusive", C.c_int32), ("nnratio", C.c_float),
                ("F12", C.c_float * 9), ("ex", C.c_float), ("ey", C.c_float), ("kp1_xy", C.c_void_p), ("kp2_xy", C.c_void_p),
                ("kp2_octave", C.c_void_p), ("scale_factor", C.c_float * 16), ("level_sigma2", C.c_float * 16)]


class Sim3Problem(C.Structure):
    _fields_ = [("n", C.c_int32), ("p1", C.c_void_p), ("p2", C.c_void_p), ("obs1", C.c_void_p), ("obs2", C.c_void_p),
                ("inv_sigma2_1", C.c_void_p), ("inv_sigma2_2", C.c_void_p), ("intr1", C.c_double * 4), ("intr2", C.c_double * 4),
                ("r12", C.c_double * 9), ("t12", C.c_double * 3), ("s12", C.c_double), ("th2", C.c_double), ("fix_scale", C.c_int32)]


class Sim3Result(C.Structure):
    _fields_ = [("r12", C.c_double * 9), ("t12", C.c_double * 3), ("s12", C.c_double), ("inlier", C.c_void_p), ("n_inliers", C.c_int32),
                ("n_its", C.c_int32 * 2), ("chi2", C.c_double * 2)]


class SearchBatch(C.Structure):
    _fields_ = [("nframes", C.c_int32), ("kp_cap", C.c_int32), ("q_cap", C.c_int32), ("d_n", C.c_void_p),
                ("d_kps_un", C.c_void_p), ("d_desc", C.c_void_p), ("d_kp_taken", C.c_void_p), ("min_x", C.c_float),
                ("min_y", C.c_float), ("inv_w", C.c_float), ("inv_h", C.c_float), ("d_m", C.c_void_p), ("d_uvr", C.c_void_p),
                ("d_level_min", C.c_void_p), ("d_level_max", C.c_void_p), ("d_qdesc", C.c_void_p), ("d_valid", C.c_void_p),
                ("d_takes", C.c_void_p)]


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("k1", C.c_float),
                ("k2", C.c_float), ("p1", C.c_float), ("p2", C.c_float), ("k3", C.c_float)]


class BaProfile(C.Structure):
    _fields_ = [("phase_ms", C.c_double * 5), ("slots", C.c_int32), ("nwin", C.c_int32), ("schur_exec_mflop", C.c_double)]


BA_PHASES = ("linearize", "schur", "solve", "update", "residuals")   # slamit_ba_profile_out.phase_ms


class PoseProblem(C.Structure):
    _fields_ = [("n", C.c_int32), ("pose", C.c_void_p), ("intr", C.c_void_p), ("xw", C.c_void_p),
                ("uv", C.c_void_p), ("inv_sigma2", C.c_void_p), ("ur", C.c_void_p), ("bf", C.c_double)]   # stereo: right-image columns (< 0: monocular) and Frame::mbf


class PoseResult(C.Structure):
    _fields_ = [("pose", C.c_void_p), ("outlier", C.c_void_p), ("n_inliers", C.c_int32),
                ("n_its", C.c_int32 * 4), ("chi2", C.c_double * 4)]


_lib = None

EXPORTS = [
    "slamit_orb_create", "slamit_orb_destroy", "slamit_orb_tables", "slamit_orb_max_keypoints",
    "slamit_orb_extract", "slamit_orb_extract_batch", "slamit_orb_extract_batch_dev", "slamit_orb_level",
    "slamit_orb_debug_candidates", "slamit_orb_debug_blurred", "slamit_orb_profile", "slamit_hamming_best2", "slamit_hamming_best2_batch_dev",
    "slamit_hamming_matrix", "slamit_distinctive_batch", "slamit_guided_search", "slamit_guided_search_workspace", "slamit_guided_search_batch_dev", "slamit_bow_search", "slamit_undistort_points", "slamit_frame_finish",
    "slamit_frame_finish_batch_dev", "slamit_ba_create", "slamit_ba_destroy", "slamit_ba_solve",
    "slamit_ba_solve_batch", "slamit_ba_profile", "slamit_ba_profile_read", "slamit_pose_optimize", "slamit_pose_optimize_batch", "slamit_sim3_optimize", "slamit_sim3_optimize_batch", "slamit_last_error", "slamit_version", "slamit_device_count", "slamit_set_device", "slamit_release_thread_scratch",
]


def lib():
    """Loads the HIP library; raises (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SlamitError(
                "libslamit_hip.so is not built: run `python -m weiner_slamit_v2_amd.build` "
                "(there is no CPU fallback)")
        # One HIP runtime per process: the PyTorch wheel bundles its own libamdhip64.so.7 /
        # libhsa-runtime64; if ours loaded /opt/rocm's copy first, torch.cuda would later find
        # "No HIP GPUs".  Importing torch first makes our DT_NEEDED resolve to the copy torch
        # already mapped (same SONAME).  Pure C/C++ users link /opt/rocm and never see this.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, i32, sz = C.c_void_p, C.c_int, C.c_size_t
        L.slamit_orb_create.argtypes = [C.POINTER(OrbParams), i32, C.POINTER(vp)]
        L.slamit_orb_destroy.argtypes = [vp]
        L.slamit_orb_destroy.restype = None
        L.slamit_orb_tables.argtypes = [vp] + [vp] * 5
        L.slamit_orb_max_keypoints.argtypes = [vp]
        L.slamit_orb_extract.argtypes = [vp, vp, sz, vp, vp, i32, vp]
        L.slamit_orb_extract_batch.argtypes = [vp, vp, sz, sz, i32, vp, vp, i32, vp]
        L.slamit_orb_extract_batch_dev.argtypes = [vp, vp, sz, sz, i32, vp, vp, i32, vp, vp]
        L.slamit_orb_level.argtypes = [vp, i32, i32, vp, sz, vp, vp]
        L.slamit_orb_debug_blurred.argtypes = [vp, i32, i32, vp, sz, vp, vp]
        L.slamit_orb_profile.argtypes = [vp, i32, vp, vp, i32]
        L.slamit_orb_debug_candidates.argtypes = [vp, i32, i32, vp, i32, vp]
        L.slamit_hamming_best2.argtypes = [vp, i32, vp, i32, vp, vp, vp]
        L.slamit_hamming_best2_batch_dev.argtypes = [vp, vp, sz, vp, vp, sz, i32, i32, vp, vp, vp, sz, i32, vp]
        L.slamit_hamming_matrix.argtypes = [vp, i32, vp, i32, vp]
        L.slamit_distinctive_batch.argtypes = [vp, vp, i32, vp, vp]
        L.slamit_guided_search.argtypes = [i32, C.POINTER(FrameView), C.POINTER(SearchQueries), C.POINTER(SearchRule),
                                           vp, vp, vp, vp, vp, vp]
        L.slamit_guided_search_workspace.argtypes = [i32, i32]
        L.slamit_guided_search_workspace.restype = sz
        L.slamit_guided_search_batch_dev.argtypes = [i32, C.POINTER(SearchBatch), C.POINTER(SearchRule), vp, vp, vp, vp, sz, vp]
        f32 = C.c_float
        L.slamit_sim3_optimize_batch.argtypes = [i32, i32, C.POINTER(Sim3Problem), C.POINTER(Sim3Result)]
        L.slamit_sim3_optimize.argtypes = [i32, C.POINTER(Sim3Problem), C.POINTER(Sim3Result)]
        L.slamit_bow_search.argtypes = [i32, vp, i32, vp, vp, i32, vp, C.POINTER(BowGroups), C.POINTER(BowRule), vp, vp, vp]
        L.slamit_undistort_points.argtypes = [i32, C.POINTER(Camera), vp, i32, vp]
        L.slamit_frame_finish.argtypes = [i32, C.POINTER(Camera), vp, i32, f32, f32, f32, f32, vp, vp, vp]
        L.slamit_frame_finish_batch_dev.argtypes = [i32, C.POINTER(Camera), vp, vp, i32, i32, f32, f32, f32, f32, vp, vp, vp, vp]
        if hasattr(L, "slamit_ba_create"):
            L.slamit_ba_create.argtypes = [i32, i32, i32, i32, i32, C.POINTER(vp)]
            L.slamit_ba_destroy.argtypes = [vp]
            L.slamit_ba_destroy.restype = None
            L.slamit_ba_solve.argtypes = [vp, C.POINTER(BaProblem), C.POINTER(BaOpts), C.POINTER(BaResult)]
            L.slamit_ba_solve_batch.argtypes = [vp, i32, C.POINTER(BaProblem), C.POINTER(BaOpts), C.POINTER(BaResult)]
            if hasattr(L, "slamit_ba_profile"):   # absent from round-2 libraries loaded through SLAMIT_LIB for A/B runs
                L.slamit_ba_profile.argtypes = [vp, i32]
                L.slamit_ba_profile_read.argtypes = [vp, C.POINTER(BaProfile)]
        if hasattr(L, "slamit_pose_optimize_batch"):
            L.slamit_pose_optimize_batch.argtypes = [i32, i32, C.POINTER(PoseProblem), C.POINTER(PoseResult)]
            L.slamit_pose_optimize.argtypes = [i32, C.POINTER(PoseProblem), C.POINTER(PoseResult)]
        L.slamit_last_error.restype = C.c_char_p
        L.slamit_version.restype = C.c_char_p
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise SlamitError("%s failed (%d): %s" % (what, rc, lib().slamit_last_error().decode()))


def device_count():
    return lib().slamit_device_count()


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class ORBextractor:
    """ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST) on one GPU.

    Frame geometry is bound lazily on the first image (the reference accepts any cv::Mat; the
    device workspace is sized per geometry and re-created if it changes)."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7,
                 device=0, max_batch=1):
        self.params = (int(nfeatures), float(scaleFactor), int(nlevels), int(iniThFAST), int(minThFAST))
        self.device = device
        self.max_batch = max_batch
        self._h = None
        self._geom = None

    # -- handle management --
    def _bind(self, width, height, batch=1):
        if self._h is not None and self._geom == (width, height) and batch <= self.max_batch:
            return
        self.close()
        self.max_batch = max(self.max_batch, batch)
        p = OrbParams(self.params[0], self.params[1], self.params[2], self.params[3], self.params[4],
                      width, height, self.max_batch)
        h = C.c_void_p()
        _check(lib().slamit_orb_create(C.byref(p), self.device, C.byref(h)), "slamit_orb_create")
        self._h = h
        self._geom = (width, height)
        self.max_keypoints = lib().slamit_orb_max_keypoints(h)

    def close(self):
        if self._h is not None:
            lib().slamit_orb_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- getters of the reference class --
    def GetLevels(self):
        return self.params[2]

    def GetScaleFactor(self):
        return self.params[1]

    def _tables(self):
        if self._h is None:
            self._bind(640, 480)
        n = self.params[2]
        out = [np.zeros(n, np.float32) for _ in range(4)] + [np.zeros(n, np.int32)]
        _check(lib().slamit_orb_tables(self._h, *[_np_ptr(a) for a in out]), "slamit_orb_tables")
        return out

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        return self._tables()[4]

    # -- operator() --
    def __call__(self, image, mask=None):
        """image: uint8 (H, W) numpy array. Returns (keypoints[KP_DTYPE], descriptors[n,32])."""
        image = np.asarray(image)
        if image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        assert image.dtype == np.uint8 and image.ndim == 2, "CV_8UC1 expected (ORBextractor.cc:1075)"
        k, d = self.extract_batch(image[None])
        return k[0], d[0]

    def extract_batch(self, frames):
        """frames: uint8 (B, H, W) host array -> lists of per-frame keypoints / descriptors."""
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        b, hh, ww = frames.shape
        self._bind(ww, hh, b)
        cap = self.max_keypoints
        kps = np.zeros((b, cap), KP_DTYPE)
        desc = np.zeros((b, cap, 32), np.uint8)
        n = np.zeros(b, np.int32)
        _check(lib().slamit_orb_extract_batch(self._h, _np_ptr(frames), ww, ww * hh, b, _np_ptr(kps), _np_ptr(desc),
                                              cap, _np_ptr(n)), "slamit_orb_extract_batch")
        return [kps[i, :n[i]].copy() for i in range(b)], [desc[i, :n[i]].copy() for i in range(b)]

    def extract_batch_dev(self, d_frames, d_kps, d_desc, d_n, stream=None):
        """torch uint8 CUDA tensor (B,H,W) -> fills d_kps (B,cap,7 float32 view), d_desc (B,cap,32), d_n (B)."""
        b, hh, ww = d_frames.shape
        self._bind(ww, hh, b)
        cap = d_kps.shape[1]
        _check(lib().slamit_orb_extract_batch_dev(
            self._h, d_frames.data_ptr(), d_frames.stride(1), d_frames.stride(0), b, d_kps.data_ptr(),
            d_desc.data_ptr(), cap, d_n.data_ptr(), stream), "slamit_orb_extract_batch_dev")

    STAGES = ("resize", "fast", "octree", "angle", "blur", "describe")

    def profile(self, enable):
        """Returns {stage: (total_ms, calls)} accumulated since the last call; sets recording."""
        ms = np.zeros(6, np.float32)
        calls = np.zeros(6, np.int32)
        _check(lib().slamit_orb_profile(self._h, int(enable), _np_ptr(ms), _np_ptr(calls), 6), "slamit_orb_profile")
        return {s: (float(ms[i]), int(calls[i])) for i, s in enumerate(self.STAGES)}

    def level(self, frame, level):
        """mvImagePyramid[level] of `frame` of the last call, padded plane (h+38, w+38)."""
        w, h = C.c_int(), C.c_int()
        _check(lib().slamit_orb_level(self._h, frame, level, None, 0, C.byref(w), C.byref(h)), "slamit_orb_level")
        out = np.zeros((h.value + 38, w.value + 38), np.uint8)
        _check(lib().slamit_orb_level(self._h, frame, level, _np_ptr(out), out.size, C.byref(w), C.byref(h)),
               "slamit_orb_level")
        return out

    def blurred(self, frame, level):
        """The blurred level (h, w) the descriptors of `frame` of the last call were sampled from."""
        w, h = C.c_int(), C.c_int()
        _check(lib().slamit_orb_debug_blurred(self._h, frame, level, None, 0, C.byref(w), C.byref(h)), "slamit_orb_debug_blurred")
        out = np.zeros((h.value, w.value), np.uint8)
        _check(lib().slamit_orb_debug_blurred(self._h, frame, level, _np_ptr(out), out.size, C.byref(w), C.byref(h)),
               "slamit_orb_debug_blurred")
        return out

    def debug_candidates(self, frame, level):
        n = C.c_int()
        _check(lib().slamit_orb_debug_candidates(self._h, frame, level, None, 0, C.byref(n)), "debug_candidates")
        out = np.zeros((max(n.value, 1), 3), np.int32)
        _check(lib().slamit_orb_debug_candidates(self._h, frame, level, _np_ptr(out), n.value, C.byref(n)),
               "debug_candidates")
        return out[:n.value]


GRID_COLS, GRID_ROWS = 64, 48   # FRAME_GRID_COLS / FRAME_GRID_ROWS (include/Frame.h:40-41)


class Frame:
    """The part of ORB_SLAM2::Frame's constructor that follows the extractor (src/Frame.cc:84-117): image bounds,
    UndistortKeyPoints, AssignFeaturesToGrid.  cam9 = fx fy cx cy k1 k2 p1 p2 k3."""

    @staticmethod
    def undistort_points(cam9, xy, device=0):
        cam = Camera(*[float(v) for v in cam9])
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        out = np.zeros_like(xy)
        _check(lib().slamit_undistort_points(device, C.byref(cam), _np_ptr(xy), len(xy), _np_ptr(out)), "slamit_undistort_points")
        return out

    @staticmethod
    def ComputeImageBounds(cam9, cols, rows, device=0):
        """(mnMinX, mnMaxX, mnMinY, mnMaxY, mfGridElementWidthInv, mfGridElementHeightInv), Frame.cc:561-590, :113-118."""
        f32 = np.float32
        if f32(cam9[4]) != 0:
            m = Frame.undistort_points(cam9, [[0, 0], [cols, 0], [0, rows], [cols, rows]], device)
            b = (min(m[0, 0], m[2, 0]), max(m[1, 0], m[3, 0]), min(m[0, 1], m[1, 1]), max(m[2, 1], m[3, 1]))
        else:
            b = (f32(0), f32(cols), f32(0), f32(rows))
        b = tuple(f32(v) for v in b)
        return b + (f32(GRID_COLS) / f32(b[1] - b[0]), f32(GRID_ROWS) / f32(b[3] - b[2]))

    @staticmethod
    def finish(cam9, kps, min_x, min_y, inv_w, inv_h, device=0):
        """UndistortKeyPoints + AssignFeaturesToGrid: (mvKeysUn, cell_start[3073], cell_items)."""
        cam = Camera(*[float(v) for v in cam9])
        kps = np.ascontiguousarray(kps)
        n = len(kps)
        un = np.zeros(max(n, 1), KP_DTYPE)
        start, items = np.zeros(GRID_COLS * GRID_ROWS + 1, np.int32), np.zeros(max(n, 1), np.int32)
        _check(lib().slamit_frame_finish(device, C.byref(cam), _np_ptr(kps), n, float(min_x), float(min_y), float(inv_w),
                                         float(inv_h), _np_ptr(un), _np_ptr(start), _np_ptr(items)), "slamit_frame_finish")
        return un[:n], start, items[:start[-1]]

    @staticmethod
    def finish_batch_dev(cam9, d_kps, d_n, min_x, min_y, inv_w, inv_h, d_kps_un, d_cell_start, d_cell_items, device=0, stream=None):
        """torch tensors in the extractor's layout: d_kps (B, cap, 7) float32 view of cv::KeyPoint records."""
        cam = Camera(*[float(v) for v in cam9])
        b, cap = d_kps.shape[0], d_kps.shape[1]
        _check(lib().slamit_frame_finish_batch_dev(device, C.byref(cam), d_kps.data_ptr(), d_n.data_ptr(), cap, b, float(min_x),
                                                   float(min_y), float(inv_w), float(inv_h), d_kps_un.data_ptr(),
                                                   d_cell_start.data_ptr(), d_cell_items.data_ptr(), stream),
               "slamit_frame_finish_batch_dev")


class ORBmatcher:
    TH_HIGH = 100      # ORBmatcher.cc:37
    TH_LOW = 50        # :38
    HISTO_LENGTH = 30  # :39

    def __init__(self, nnratio=0.6, checkOri=True):
        self.mfNNratio = nnratio
        self.mbCheckOrientation = checkOri

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.ascontiguousarray(a, np.uint8).reshape(1, 32)
        b = np.ascontiguousarray(b, np.uint8).reshape(1, 32)
        return int(ORBmatcher.distance_matrix(a, b)[0, 0])

    @staticmethod
    def distance_matrix(q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        out = np.zeros((len(q), len(t)), np.uint16)
        _check(lib().slamit_hamming_matrix(_np_ptr(q), len(q), _np_ptr(t), len(t), _np_ptr(out)), "slamit_hamming_matrix")
        return out

    @staticmethod
    def best2(q, t):
        q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
        t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
        idx, best, second = (np.zeros(len(q), np.int32) for _ in range(3))
        _check(lib().slamit_hamming_best2(_np_ptr(q), len(q), _np_ptr(t), len(t), _np_ptr(idx), _np_ptr(best),
                                          _np_ptr(second)), "slamit_hamming_best2")
        return idx, best, second

    @staticmethod
    def distinctive(desc, offsets):
        """MapPoint::ComputeDistinctiveDescriptors, batched: returns (best row per point, its median)."""
        desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        offsets = np.ascontiguousarray(offsets, np.int32)
        n = len(offsets) - 1
        idx, med = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        _check(lib().slamit_distinctive_batch(_np_ptr(desc), _np_ptr(offsets), n, _np_ptr(idx), _np_ptr(med)),
               "slamit_distinctive_batch")
        return idx[:n], med[:n]

    @staticmethod
    def guided_search(frame, queries, th_dist=100, use_ratio=True, nnratio=0.8, device=0, chi2_gate=0.0, inv_level_sigma2=None, mode=0):
        """The loop body of ORBmatcher::SearchByProjection (ORBmatcher.cc:47-131, :1332-1474) for all
        queries in order: window query over the frame grid, best/second Hamming, accept, mark taken.
        frame: dict kp_xy (n,2) f32, kp_octave (n) i32, desc (n,32) u8, kp_taken (n) u8, min_x, min_y,
        inv_w, inv_h.  queries: dict uvr (m,3) f32, level_min, level_max (m) i32, desc (m,32) u8,
        optional valid / takes (m) u8.  Returns (match_kp[m], nmatches, out4[m,4])."""
        f = dict(kp_xy=np.ascontiguousarray(frame["kp_xy"], np.float32).reshape(-1, 2),
                 kp_octave=np.ascontiguousarray(frame["kp_octave"], np.int32),
                 desc=np.ascontiguousarray(frame["desc"], np.uint8).reshape(-1, 32),
                 kp_taken=np.ascontiguousarray(frame["kp_taken"], np.uint8))
        uvr = np.ascontiguousarray(queries["uvr"], np.float32).reshape(-1, 3)
        m, n = len(uvr), len(f["kp_xy"])
        q = dict(level_min=np.ascontiguousarray(queries["level_min"], np.int32),
                 level_max=np.ascontiguousarray(queries["level_max"], np.int32),
                 desc=np.ascontiguousarray(queries["desc"], np.uint8).reshape(-1, 32),
                 valid=np.ascontiguousarray(queries.get("valid", np.ones(m)), np.uint8),
                 takes=np.ascontiguousarray(queries.get("takes", np.ones(m)), np.uint8))
        for k, a in list(f.items())[1:]:
            if len(a) != n:
                raise SlamitError("guided_search: frame[%s] has %d rows, expected %d" % (k, len(a), n))
        for k, a in q.items():
            if len(a) != m:
                raise SlamitError("guided_search: queries[%s] has %d rows, expected %d" % (k, len(a), m))
        fv = FrameView(n, _np_ptr(f["kp_xy"]), _np_ptr(f["kp_octave"]), _np_ptr(f["desc"]), _np_ptr(f["kp_taken"]),
                       frame["min_x"], frame["min_y"], frame["inv_w"], frame["inv_h"])
        sq = SearchQueries(m, _np_ptr(uvr), _np_ptr(q["level_min"]), _np_ptr(q["level_max"]), _np_ptr(q["desc"]),
                           _np_ptr(q["valid"]), _np_ptr(q["takes"]))
        rule = _search_rule(th_dist, use_ratio, nnratio, chi2_gate, inv_level_sigma2, mode)
        match = np.full(max(m, 1), -1, np.int32)
        out4 = np.zeros((4, max(m, 1)), np.int32)
        nm = C.c_int32(0)
        _check(lib().slamit_guided_search(device, C.byref(fv), C.byref(sq), C.byref(rule), _np_ptr(match), C.byref(nm),
                                          _np_ptr(out4[0]), _np_ptr(out4[1]), _np_ptr(out4[2]), _np_ptr(out4[3])),
               "slamit_guided_search")
        return match[:m], nm.value, out4[:, :m].T.copy()

    @staticmethod
    def guided_search_batch_dev(t, bounds, th_dist=100, use_ratio=True, nnratio=0.8, device=0, stream=None, chi2_gate=0.0,
                                inv_level_sigma2=None):
        """Batched device form.  t: dict of torch CUDA tensors n (B) i32, kps_un (B, kp_cap, 7) f32 view of cv::KeyPoint,
        desc (B, kp_cap, 32) u8, kp_taken (B, kp_cap) u8, m (B) i32, uvr (B, q_cap, 3) f32, level_min / level_max
        (B, q_cap) i32, qdesc (B, q_cap, 32) u8, valid / takes (B, q_cap) u8, match_kp (B, q_cap) i32, nmatches (B) i32,
        out4 (B, q_cap, 4) i32 or None, workspace (bytes,) u8.  bounds = (min_x, min_y, inv_w, inv_h)."""
        b, kp_cap, q_cap = t["kps_un"].shape[0], t["kps_un"].shape[1], t["uvr"].shape[1]
        sb = SearchBatch(b, kp_cap, q_cap, t["n"].data_ptr(), t["kps_un"].data_ptr(), t["desc"].data_ptr(), t["kp_taken"].data_ptr(),
                         float(bounds[0]), float(bounds[1]), float(bounds[2]), float(bounds[3]), t["m"].data_ptr(), t["uvr"].data_ptr(),
                         t["level_min"].data_ptr(), t["level_max"].data_ptr(), t["qdesc"].data_ptr(), t["valid"].data_ptr(),
                         t["takes"].data_ptr())
        rule = _search_rule(th_dist, use_ratio, nnratio, chi2_gate, inv_level_sigma2)
        out4 = t.get("out4")
        _check(lib().slamit_guided_search_batch_dev(device, C.byref(sb), C.byref(rule), t["match_kp"].data_ptr(), t["nmatches"].data_ptr(),
                                                    out4.data_ptr() if out4 is not None else None, t["workspace"].data_ptr(),
                                                    t["workspace"].numel(), stream), "slamit_guided_search_batch_dev")

    @staticmethod
    def guided_search_workspace(nframes, q_cap):
        return int(lib().slamit_guided_search_workspace(nframes, q_cap))

    @staticmethod
    def search_for_initialization(f1, prev_xy, f2, window=100, nnratio=0.9, th_low=50, device=0):
        """ORBmatcher::SearchForInitialization's matching loop (ORBmatcher.cc:409-474) as guided-search mode 1: queries =
        F1's level-0 keypoints at vbPrevMatched, window `window`, level [0, 0].  Returns (vnMatches12, nmatches, accepted-at-turn)."""
        o1 = np.ascontiguousarray(f1["kp_octave"], np.int32)
        n1 = len(o1)
        pv = np.ascontiguousarray(prev_xy, np.float32).reshape(-1, 2)
        q = dict(uvr=np.concatenate([pv, np.full((n1, 1), float(window), np.float32)], 1), level_min=np.zeros(n1, np.int32),
                 level_max=np.zeros(n1, np.int32), desc=f1["desc"], valid=(o1 <= 0).astype(np.uint8), takes=np.zeros(n1, np.uint8))
        frame = dict(f2, kp_taken=np.zeros(len(np.asarray(f2["kp_octave"])), np.uint8))
        m12, nm, out4 = ORBmatcher.guided_search(frame, q, th_low, False, nnratio, device=device, mode=1)
        return m12, nm, out4[:, 1].copy()   # the level slot carries the keypoint accepted at the query's own turn

    @staticmethod
    def bow_search(side1, side2, groups, mode=0, th=50, th_inclusive=True, nnratio=0.6, epi=None, device=0):
        """The matching loops of SearchByBoW (mode 0; ORBmatcher.cc:161-290, 526-657) and SearchForTriangulation (mode 1;
        :659-826) over vocabulary-node groups.  side1 / side2: dicts with desc (n, 32), optional valid (n) and, for mode 1,
        kp_xy (n, 2) (+ kp_octave on side 2); groups: dict q_ptr, q_idx, c_ptr, c_idx (CSR per common node); epi (mode 1):
        dict F12 (9, row-major), ex, ey, scale_factor (16), level_sigma2 (16).  Returns (match12, dist12, nmatches)
        before the rotation-histogram filter."""
        d1 = np.ascontiguousarray(side1["desc"], np.uint8).reshape(-1, 32)
        d2 = np.ascontiguousarray(side2["desc"], np.uint8).reshape(-1, 32)
        n1, n2 = len(d1), len(d2)
        v1 = None if side1.get("valid") is None else np.ascontiguousarray(side1["valid"], np.uint8)
        v2 = None if side2.get("valid") is None else np.ascontiguousarray(side2["valid"], np.uint8)
        qp, qi, cp, ci = (np.ascontiguousarray(groups[k], np.int32) for k in ("q_ptr", "q_idx", "c_ptr", "c_idx"))
        g = BowGroups(len(qp) - 1, qp.ctypes.data, qi.ctypes.data, cp.ctypes.data, ci.ctypes.data)
        rule = BowRule()
        rule.mode, rule.th, rule.th_inclusive, rule.nnratio = int(mode), int(th), int(bool(th_inclusive)), float(nnratio)
        keep = []
        if mode == 1:
            k1 = np.ascontiguousarray(side1["kp_xy"], np.float32).reshape(-1, 2)
            k2 = np.ascontiguousarray(side2["kp_xy"], np.float32).reshape(-1, 2)
            o2 = np.ascontiguousarray(side2["kp_octave"], np.int32)
            keep = [k1, k2, o2]
            rule.F12 = (C.c_float * 9)(*[float(v) for v in np.asarray(epi["F12"], np.float32).reshape(9)])
            rule.ex, rule.ey = float(np.float32(epi["ex"])), float(np.float32(epi["ey"]))
            rule.kp1_xy, rule.kp2_xy, rule.kp2_octave = k1.ctypes.data, k2.ctypes.data, o2.ctypes.data
            rule.scale_factor = (C.c_float * 16)(*[float(v) for v in list(epi["scale_factor"])[:16] + [1.0] * (16 - len(epi["scale_factor"]))])
            rule.level_sigma2 = (C.c_float * 16)(*[float(v) for v in list(epi["level_sigma2"])[:16] + [1.0] * (16 - len(epi["level_sigma2"]))])
        m12, dd = np.full(max(n1, 1), -1, np.int32), np.full(max(n1, 1), 256, np.int32)
        nm = C.c_int32(0)
        _check(lib().slamit_bow_search(device, _np_ptr(d1), n1, _np_ptr(v1) if v1 is not None else None, _np_ptr(d2), n2,
                                       _np_ptr(v2) if v2 is not None else None, C.byref(g), C.byref(rule), _np_ptr(m12),
                                       _np_ptr(dd), C.byref(nm)), "slamit_bow_search")
        del keep
        return m12[:n1], dd[:n1], nm.value

    def match(self, q, t, th=None):
        """All-pairs match with the reference's acceptance rule: best <= th and best < nnratio*second
        (ORBmatcher.cc:1430-1436 style). Returns query indices and their matched train indices."""
        th = self.TH_LOW if th is None else th
        idx, best, second = self.best2(q, t)
        ok = (best <= th) & (best.astype(np.float32) < np.float32(self.mfNNratio) * second.astype(np.float32))
        return np.nonzero(ok)[0], idx[ok]

    @staticmethod
    def best2_batch_dev(d_q, d_nq, d_t, d_nt, d_idx, d_best, d_second, max_n, device=0, stream=None):
        """torch tensors: d_q/d_t (P, cap, 32) uint8, d_nq/d_nt (P) int32, outputs (P, cap) int32."""
        p = d_q.shape[0]
        _check(lib().slamit_hamming_best2_batch_dev(
            d_q.data_ptr(), d_nq.data_ptr(), d_q.stride(0), d_t.data_ptr(), d_nt.data_ptr(), d_t.stride(0), p, max_n,
            d_idx.data_ptr(), d_best.data_ptr(), d_second.data_ptr(), d_idx.stride(0), device, stream),
            "slamit_hamming_best2_batch_dev")


def _ba_problem(arrs):
    keep = {}
    for k, dt in (("kf_pose", np.float64), ("kf_fixed", np.uint8), ("kf_intr", np.float64), ("pt_xyz", np.float64),
                  ("edge_kf", np.int32), ("edge_pt", np.int32), ("edge_uv", np.float64), ("edge_inv_sigma2", np.float64)):
        keep[k] = np.ascontiguousarray(arrs[k], dtype=dt)
    stereo = arrs.get("edge_ur") is not None
    if stereo:   # right-image columns (negative: a monocular edge) and the keyframes' baseline x fx
        keep["edge_ur"] = np.ascontiguousarray(arrs["edge_ur"], dtype=np.float64)
        keep["kf_bf"] = np.ascontiguousarray(arrs["kf_bf"], dtype=np.float64)
        if len(keep["edge_ur"]) != len(keep["edge_kf"]) or len(keep["kf_bf"]) != len(keep["kf_fixed"]):
            raise SlamitError("edge_ur / kf_bf do not match the window's edges / keyframes")
    p = BaProblem(len(keep["kf_fixed"]), len(keep["pt_xyz"]), len(keep["edge_kf"]),
                  *[keep[k].ctypes.data for k in ("kf_pose", "kf_fixed", "kf_intr", "pt_xyz", "edge_kf", "edge_pt",
                                                   "edge_uv", "edge_inv_sigma2")],
                  keep["edge_ur"].ctypes.data if stereo else None, keep["kf_bf"].ctypes.data if stereo else None)
    return p, keep


HUBER_MONO = float(np.float32(np.sqrt(5.991)))  # Optimizer.cc:569 stores sqrt(5.991) in a float
HUBER_STEREO = float(np.float32(np.sqrt(7.815)))  # Optimizer.cc:570


class Optimizer:
    """Optimizer::LocalBundleAdjustment on POD inputs (the KeyFrame/MapPoint gathering of
    Optimizer.cc:456-504 stays with the caller)."""

    def __init__(self, max_kf=64, max_pt=4096, max_edge=262144, max_batch=1, device=0):
        h = C.c_void_p()
        _check(lib().slamit_ba_create(max_kf, max_pt, max_edge, max_batch, device, C.byref(h)), "slamit_ba_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None) is not None:
            lib().slamit_ba_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def profile(self, on=True):
        """Per-phase timing of the following solves (slamit_ba_profile): events between the phases of every LM slot."""
        _check(lib().slamit_ba_profile(self._h, 1 if on else 0), "slamit_ba_profile")

    def profile_read(self):
        """Phase sums of the last profiled solve: {"phase_ms": {linearize, schur, solve, update, residuals}, "slots", "nwin",
        "schur_exec_mflop"} — the analogue of g2o's G2OBatchStatistics."""
        o = BaProfile()
        _check(lib().slamit_ba_profile_read(self._h, C.byref(o)), "slamit_ba_profile_read")
        return {"phase_ms": dict(zip(BA_PHASES, [float(v) for v in o.phase_ms])), "slots": int(o.slots), "nwin": int(o.nwin),
                "schur_exec_mflop": float(o.schur_exec_mflop)}

    @staticmethod
    def _result(n_kf, n_pt, n_e):
        # (np.empty: the call writes every element of every output or fails; zero-filling 210 KB per window was 4 ms of a 17-ms batch of 64)
        out = {"kf_pose": np.empty((n_kf, 12)), "pt_xyz": np.empty((n_pt, 3)), "edge_chi2": np.empty(n_e),
               "edge_outlier": np.empty(n_e, np.uint8), "edge_stage1_outlier": np.empty(n_e, np.uint8)}
        st = BaStats()
        r = BaResult(out["kf_pose"].ctypes.data, out["pt_xyz"].ctypes.data, out["edge_chi2"].ctypes.data,
                     out["edge_outlier"].ctypes.data, out["edge_stage1_outlier"].ctypes.data, C.addressof(st))
        return r, out, st

    @staticmethod
    def _stats(st):
        n = list(st.n_its)
        return {"n_its": n, "chi2": [st.chi2[s][:n[s]] for s in range(2)],
                "lambda": [st.lambda_[s][:n[s]] for s in range(2)],
                "trials": [st.trials[s][:n[s]] for s in range(2)], "chi2_init": list(st.chi2_init)}

    def LocalBundleAdjustment(self, problem, its_robust=5, its_final=10, huber_delta=HUBER_MONO, chi2_gate=5.991,
                              stop=None, huber_delta_stereo=HUBER_STEREO, chi2_gate_stereo=7.815):
        p, keep = _ba_problem(problem)
        o = BaOpts(its_robust, its_final, huber_delta, chi2_gate, stop.ctypes.data if stop is not None else None,
                   huber_delta_stereo, chi2_gate_stereo)
        r, out, st = self._result(p.n_kf, p.n_pt, p.n_edge)
        _check(lib().slamit_ba_solve(self._h, C.byref(p), C.byref(o), C.byref(r)), "slamit_ba_solve")
        out["stats"] = self._stats(st)
        return out

    @staticmethod
    def OptimizeSim3(problems, device=0):
        """Optimizer::OptimizeSim3 for one problem dict or a list of them (synth.synth_sim3 layout: p1, p2, obs1, obs2,
        inv_sigma2_1, inv_sigma2_2, intr1, intr2, r12, t12, s12, th2, fix_scale).  Returns dict(s) with r12 (3, 3), t12, s12,
        inlier flags, n_inliers, n_its[2], chi2[2]."""
        single = isinstance(problems, dict)
        plist = [problems] if single else list(problems)
        m = len(plist)
        P = (Sim3Problem * m)()
        R = (Sim3Result * m)()
        keep, flags = [], []
        for i, pr in enumerate(plist):
            k = {key: np.ascontiguousarray(pr[key], np.float64) for key in ("p1", "p2", "obs1", "obs2", "inv_sigma2_1", "inv_sigma2_2")}
            q = P[i]
            q.n = len(k["inv_sigma2_1"])
            for key, a in k.items():
                setattr(q, key, a.ctypes.data)
            q.intr1 = (C.c_double * 4)(*[float(v) for v in pr["intr1"]])
            q.intr2 = (C.c_double * 4)(*[float(v) for v in pr["intr2"]])
            q.r12 = (C.c_double * 9)(*[float(v) for v in np.asarray(pr["r12"]).reshape(9)])
            q.t12 = (C.c_double * 3)(*[float(v) for v in pr["t12"]])
            q.s12, q.th2, q.fix_scale = float(pr["s12"]), float(pr["th2"]), int(pr["fix_scale"])
            fl = np.zeros(max(q.n, 1), np.uint8)
            R[i].inlier = fl.ctypes.data
            keep.append(k)
            flags.append(fl)
        _check(lib().slamit_sim3_optimize_batch(device, m, P, R), "slamit_sim3_optimize_batch")
        outs = [{"r12": np.array(R[i].r12[:]).reshape(3, 3), "t12": np.array(R[i].t12[:]), "s12": R[i].s12, "inlier": flags[i][:P[i].n].copy(),
                 "n_inliers": R[i].n_inliers, "n_its": list(R[i].n_its), "chi2": list(R[i].chi2)} for i in range(m)]
        del keep
        return outs[0] if single else outs

    @staticmethod
    def PoseOptimization(problems, device=0):
        """Optimizer::PoseOptimization for one problem dict or a list of them (synth.synth_pose layout):
        returns dict(s) with pose (12), outlier flags, n_inliers, n_its[4], chi2[4]."""
        single = isinstance(problems, dict)
        plist = [problems] if single else list(problems)
        n = len(plist)
        P = (PoseProblem * n)()
        R = (PoseResult * n)()
        keep, outs = [], []
        for i, pr in enumerate(plist):
            k = {key: np.ascontiguousarray(pr[key], np.float64) for key in ("pose", "intr", "xw", "uv", "inv_sigma2")}
            m = len(k["inv_sigma2"])
            if pr.get("ur") is not None:
                k["ur"] = np.ascontiguousarray(pr["ur"], np.float64)
                if len(k["ur"]) != m:
                    raise SlamitError("ur does not match the correspondences")
            P[i] = PoseProblem(m, *[k[key].ctypes.data for key in ("pose", "intr", "xw", "uv", "inv_sigma2")],
                               k["ur"].ctypes.data if "ur" in k else None, float(pr.get("bf", 0.0)))
            o = {"pose": np.zeros(12), "outlier": np.zeros(max(m, 1), np.uint8)}
            R[i] = PoseResult(o["pose"].ctypes.data, o["outlier"].ctypes.data, 0)
            keep.append(k)
            outs.append((o, m))
        _check(lib().slamit_pose_optimize_batch(device, n, P, R), "slamit_pose_optimize_batch")
        res = [{"pose": o["pose"], "outlier": o["outlier"][:m].copy(), "n_inliers": R[i].n_inliers,
                "n_its": list(R[i].n_its), "chi2": list(R[i].chi2)} for i, (o, m) in enumerate(outs)]
        return res[0] if single else res

    def LocalBundleAdjustmentBatch(self, problems, its_robust=5, its_final=10, huber_delta=HUBER_MONO,
                                   chi2_gate=5.991, huber_delta_stereo=HUBER_STEREO, chi2_gate_stereo=7.815):
        n = len(problems)
        P = (BaProblem * n)()
        R = (BaResult * n)()
        keeps, outs, sts = [], [], []
        for i, prob in enumerate(problems):
            P[i], keep = _ba_problem(prob)
            keeps.append(keep)
            R[i], out, st = self._result(P[i].n_kf, P[i].n_pt, P[i].n_edge)
            outs.append(out)
            sts.append(st)
        o = BaOpts(its_robust, its_final, huber_delta, chi2_gate, None, huber_delta_stereo, chi2_gate_stereo)
        _check(lib().slamit_ba_solve_batch(self._h, n, P, C.byref(o), R), "slamit_ba_solve_batch")
        for out, st in zip(outs, sts):
            out["stats"] = self._stats(st)
        return outs
