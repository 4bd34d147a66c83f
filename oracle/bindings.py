"""ctypes bindings of the CPU oracle.  TEST INFRASTRUCTURE ONLY.

Import this module only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under weiner_slamit_v2_amd/ may import it (tests/test_layout.py checks that).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")]
)
assert KP_DTYPE.itemsize == 28

_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_f32p = C.POINTER(C.c_float)


def build(force=False):
    """Compile the oracle libraries with the committed Makefile (g++, seconds)."""
    need = force or not all(
        os.path.exists(os.path.join(HERE, n)) for n in ("liborb_oracle.so", "libba_oracle.so")
    )
    if need:
        subprocess.check_call(["make", "-s", "-C", HERE, "-f", os.path.join(HERE, "Makefile"), "all"])


def _ptr(a, t=_u8p):
    return a.ctypes.data_as(t)


_orb = None
_orb_name = "liborb_oracle.so"


def use_native_comparator():
    """bench.py's cpu_baseline only: (re)builds liborb_oracle_native.so (-O3 -march=native) ON THIS HOST and makes the
    ORB oracle load it from now on.  Timing comparator, never the parity checker (tests keep the strict -O2 build).
    Returns False (and keeps the strict build) when it cannot be built."""
    global _orb, _orb_name
    try:
        path = os.path.join(HERE, "liborb_oracle_native.so")
        if os.path.exists(path):
            os.remove(path)        # a copy built on another host may use instructions this one lacks
        subprocess.check_call(["make", "-s", "-C", HERE, "-f", os.path.join(HERE, "Makefile"), "native"])
        _orb, _orb_name = None, "liborb_oracle_native.so"
        return True
    except Exception:
        return False


def orb_lib():
    global _orb
    if _orb is None:
        build()
        L = C.CDLL(os.path.join(HERE, _orb_name))
        L.orb_oracle_create.restype = C.c_void_p
        L.orb_oracle_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orb_oracle_destroy.argtypes = [C.c_void_p]
        L.orb_oracle_tables.argtypes = [C.c_void_p, _f32p, _f32p, _f32p, _f32p, _i32p, _i32p]
        L.orb_oracle_extract.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int, C.c_void_p, _u8p, C.c_int]
        L.orb_oracle_level_size.argtypes = [C.c_void_p, C.c_int, _i32p, _i32p]
        L.orb_oracle_level.argtypes = [C.c_void_p, C.c_int, _u8p]
        L.orb_oracle_blurred.argtypes = [C.c_void_p, C.c_int, _u8p]
        L.orb_oracle_candidates.argtypes = [C.c_void_p, C.c_int, _i32p, C.c_int]
        L.orb_oracle_level_kps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orb_oracle_resize.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int, C.c_int, C.c_int]
        L.orb_oracle_blur.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p, C.c_int]
        L.orb_oracle_gauss_taps.argtypes = [_i32p]
        L.orb_oracle_fast_atan2.restype = C.c_float
        L.orb_oracle_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orb_oracle_round.argtypes = [C.c_double]
        L.orb_oracle_fast.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, C.c_int, _i32p, C.c_int]
        L.orb_oracle_octree.argtypes = [_f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32p, C.c_int]
        L.orb_oracle_descriptor.argtypes = [C.c_float, C.c_float, C.c_float, _u8p, C.c_int, _u8p]
        L.orb_oracle_cossin.argtypes = [C.c_float, _f32p, _f32p]
        L.orb_oracle_distance.argtypes = [_u8p, _u8p]
        L.orb_oracle_best2.argtypes = [_u8p, C.c_int, _u8p, C.c_int, _i32p, _i32p, _i32p]
        L.orb_oracle_matrix.argtypes = [_u8p, C.c_int, _u8p, C.c_int, C.POINTER(C.c_uint16)]
        L.orb_oracle_distinctive.argtypes = [_u8p, _i32p, C.c_int, _i32p, _i32p]
        L.orb_oracle_guided_search.argtypes = [C.c_int, _f32p, _i32p, _u8p, _u8p, C.c_float, C.c_float, C.c_float, C.c_float,
                                               C.c_int, _f32p, _i32p, _i32p, _u8p, _u8p, _u8p, C.c_int, C.c_int, C.c_float,
                                               C.c_float, _f32p, _i32p, _i32p]
        L.orb_oracle_search_init.argtypes = [C.c_int, _i32p, _u8p, _f32p, C.c_int, _f32p, _i32p, _u8p, C.c_float, C.c_float, C.c_float,
                                             C.c_float, C.c_int, C.c_float, C.c_int, _i32p, _i32p]
        L.orb_oracle_undistort.argtypes = [_f32p, _f32p, C.c_int, _f32p]
        L.orb_oracle_bow_search.argtypes = [C.c_int, _u8p, C.c_int, _u8p, _u8p, C.c_int, _u8p, C.c_int, _i32p, _i32p, _i32p, _i32p,
                                            C.c_int, C.c_int, C.c_float, _f32p, C.c_float, C.c_float, _f32p, _f32p, _i32p, _f32p,
                                            _f32p, _i32p, _i32p]
        L.orb_oracle_frame_finish.argtypes = [_f32p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p,
                                              _i32p, _i32p]
        _orb = L
    return _orb


class OrbOracle:
    """ORBextractor restatement (oracle/orb_oracle.cc)."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = orb_lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = self.L.orb_oracle_create(nfeatures, scale_factor, nlevels, ini_th, min_th)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orb_oracle_destroy(self.h)
            self.h = None

    def tables(self):
        n = self.nlevels
        sc, isc, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        per = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        self.L.orb_oracle_tables(self.h, _ptr(sc, _f32p), _ptr(isc, _f32p), _ptr(s2, _f32p), _ptr(is2, _f32p),
                                 _ptr(per, _i32p), _ptr(umax, _i32p))
        return {"scale": sc, "inv_scale": isc, "sigma2": s2, "inv_sigma2": is2, "per_level": per, "umax": umax}

    def extract(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape if img.size else (0, 0)
        cap = self.nfeatures + 3 * self.nlevels + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.L.orb_oracle_extract(self.h, _ptr(img), w, h, w, kps.ctypes.data, _ptr(desc), cap)
        assert n >= 0, "oracle keypoint capacity"
        return kps[:n].copy(), desc[:n].copy()

    def level_size(self, level):
        w, h = C.c_int32(), C.c_int32()
        assert self.L.orb_oracle_level_size(self.h, level, C.byref(w), C.byref(h)) == 0
        return w.value, h.value

    def level(self, level):
        w, h = self.level_size(level)
        out = np.zeros((h + 38, w + 38), np.uint8)
        assert self.L.orb_oracle_level(self.h, level, _ptr(out)) == 0
        return out

    def blurred(self, level):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        if self.L.orb_oracle_blurred(self.h, level, _ptr(out)) != 0:
            return None
        return out

    def candidates(self, level):
        n = self.L.orb_oracle_candidates(self.h, level, None, 0)
        out = np.zeros((max(n, 1), 3), np.int32)
        self.L.orb_oracle_candidates(self.h, level, _ptr(out, _i32p), n)
        return out[:n]

    def level_kps(self, level):
        n = self.L.orb_oracle_level_kps(self.h, level, None, 0)
        out = np.zeros(max(n, 1), KP_DTYPE)
        self.L.orb_oracle_level_kps(self.h, level, out.ctypes.data, n)
        return out[:n]


def resize(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    orb_lib().orb_oracle_resize(_ptr(src), src.shape[1], src.shape[0], src.shape[1], _ptr(dst), dw, dh, dw)
    return dst


def blur(src):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    orb_lib().orb_oracle_blur(_ptr(src), src.shape[1], src.shape[0], src.shape[1], _ptr(dst), src.shape[1])
    return dst


def gauss_taps():
    t = np.zeros(7, np.int32)
    orb_lib().orb_oracle_gauss_taps(_ptr(t, _i32p))
    return t


def fast_atan2(y, x):
    return orb_lib().orb_oracle_fast_atan2(float(y), float(x))


def cv_round(v):
    return orb_lib().orb_oracle_round(float(v))


def fast(img, threshold):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    out = np.zeros((max(cap, 1), 3), np.int32)
    n = orb_lib().orb_oracle_fast(_ptr(img), img.shape[1], img.shape[0], img.shape[1], threshold, _ptr(out, _i32p), cap)
    return out[:n]


def octree(xyr, min_x, max_x, min_y, max_y, n_target):
    xyr = np.ascontiguousarray(xyr, np.float32)
    cap = n_target + 8 + len(xyr)
    out = np.zeros((cap, 3), np.float32)
    n = orb_lib().orb_oracle_octree(_ptr(xyr, _f32p), len(xyr), min_x, max_x, min_y, max_y, n_target,
                                    _ptr(out, _f32p), cap)
    return out[:n]


def descriptor(img, kx, ky, angle_deg):
    img = np.ascontiguousarray(img, np.uint8)
    d = np.zeros(32, np.uint8)
    orb_lib().orb_oracle_descriptor(float(kx), float(ky), float(angle_deg), _ptr(img), img.shape[1], _ptr(d))
    return d


def cossin(angle_rad):
    c, s = C.c_float(), C.c_float()
    orb_lib().orb_oracle_cossin(float(angle_rad), C.byref(c), C.byref(s))
    return c.value, s.value


def distance(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return orb_lib().orb_oracle_distance(_ptr(a), _ptr(b))


def best2(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    idx, b, s = (np.zeros(len(q), np.int32) for _ in range(3))
    orb_lib().orb_oracle_best2(_ptr(q), len(q), _ptr(t), len(t), _ptr(idx, _i32p), _ptr(b, _i32p), _ptr(s, _i32p))
    return idx, b, s


def distinctive(desc, offsets):
    """MapPoint::ComputeDistinctiveDescriptors for a batch of points (rows offsets[p]:offsets[p+1])."""
    desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    n = len(offsets) - 1
    idx, med = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
    orb_lib().orb_oracle_distinctive(_ptr(desc), _ptr(offsets, _i32p), n, _ptr(idx, _i32p), _ptr(med, _i32p))
    return idx[:n], med[:n]


def guided_search(frame, queries, th_dist=100, use_ratio=True, nnratio=0.8, chi2_gate=0.0, inv_level_sigma2=None):
    """Frame grid + GetFeaturesInArea + the SearchByProjection loop (see weiner_slamit_v2_amd.api.guided_search
    for the dict layouts).  Returns (match_kp, nmatches, out4[m,4] = best dist/level, second dist/level)."""
    f, q = normalize_search(frame, queries)
    n, m = len(f["kp_xy"]), len(q["uvr"])
    match = np.full(max(m, 1), -1, np.int32)
    out4 = np.zeros((max(m, 1), 4), np.int32)
    nm = orb_lib().orb_oracle_guided_search(
        n, _ptr(f["kp_xy"], _f32p), _ptr(f["kp_octave"], _i32p), _ptr(f["desc"]), _ptr(f["kp_taken"]),
        f["min_x"], f["min_y"], f["inv_w"], f["inv_h"], m, _ptr(q["uvr"], _f32p), _ptr(q["level_min"], _i32p),
        _ptr(q["level_max"], _i32p), _ptr(q["desc"]), _ptr(q["valid"]), _ptr(q["takes"]), int(th_dist), int(bool(use_ratio)),
        float(np.float32(nnratio)), float(np.float32(chi2_gate)), _ptr(_sig16(inv_level_sigma2), _f32p), _ptr(match, _i32p),
        _ptr(out4, _i32p))
    return match[:m], nm, out4[:m]


def _sig16(v):
    out = np.ones(16, np.float32)
    if v is not None:
        v = np.asarray(v, np.float32)
        out[:len(v)] = v[:16]
    return out


def normalize_search(frame, queries):
    f = dict(kp_xy=np.ascontiguousarray(frame["kp_xy"], np.float32).reshape(-1, 2),
             kp_octave=np.ascontiguousarray(frame["kp_octave"], np.int32),
             desc=np.ascontiguousarray(frame["desc"], np.uint8).reshape(-1, 32),
             kp_taken=np.ascontiguousarray(frame["kp_taken"], np.uint8))
    for k in ("min_x", "min_y", "inv_w", "inv_h"):
        f[k] = float(np.float32(frame[k]))
    m = len(np.asarray(queries["uvr"]).reshape(-1, 3))
    q = dict(uvr=np.ascontiguousarray(queries["uvr"], np.float32).reshape(-1, 3),
             level_min=np.ascontiguousarray(queries["level_min"], np.int32),
             level_max=np.ascontiguousarray(queries["level_max"], np.int32),
             desc=np.ascontiguousarray(queries["desc"], np.uint8).reshape(-1, 32),
             valid=np.ascontiguousarray(queries.get("valid", np.ones(m)), np.uint8),
             takes=np.ascontiguousarray(queries.get("takes", np.ones(m)), np.uint8))
    return f, q


def search_for_initialization(f1, prev_xy, f2, window=100, nnratio=0.9, th_low=50):
    """ORBmatcher::SearchForInitialization's matching loop: f1 / f2 dicts with kp_octave, desc (f2 also kp_xy, min_x, min_y,
    inv_w, inv_h); prev_xy (n1, 2) = vbPrevMatched.  Returns (vnMatches12, nmatches, accepted-at-turn) before the
    rotation-histogram filter."""
    o1 = np.ascontiguousarray(f1["kp_octave"], np.int32)
    d1 = np.ascontiguousarray(f1["desc"], np.uint8).reshape(-1, 32)
    pv = np.ascontiguousarray(prev_xy, np.float32).reshape(-1, 2)
    x2 = np.ascontiguousarray(f2["kp_xy"], np.float32).reshape(-1, 2)
    o2 = np.ascontiguousarray(f2["kp_octave"], np.int32)
    d2 = np.ascontiguousarray(f2["desc"], np.uint8).reshape(-1, 32)
    m12 = np.full(max(len(o1), 1), -1, np.int32)
    acc = np.full(max(len(o1), 1), -1, np.int32)
    nm = orb_lib().orb_oracle_search_init(len(o1), _ptr(o1, _i32p), _ptr(d1), _ptr(pv, _f32p), len(o2), _ptr(x2, _f32p), _ptr(o2, _i32p),
                                          _ptr(d2), float(np.float32(f2["min_x"])), float(np.float32(f2["min_y"])),
                                          float(np.float32(f2["inv_w"])), float(np.float32(f2["inv_h"])), int(window),
                                          float(np.float32(nnratio)), int(th_low), _ptr(m12, _i32p), _ptr(acc, _i32p))
    return m12[:len(o1)], nm, acc[:len(o1)]


def bow_search(side1, side2, groups, mode=0, th=50, th_inclusive=True, nnratio=0.6, epi=None):
    """The BoW drivers' matching loops (orb_oracle_bow_search).  side1 / side2: dicts with desc (n, 32), optional valid
    (n) and, for mode 1, kp_xy (n, 2) (+ kp_octave on side 2); groups: dict q_ptr, q_idx, c_ptr, c_idx; epi (mode 1):
    dict F12 (9), ex, ey, scale_factor (16), level_sigma2 (16).  Returns (match12, dist12, nmatches)."""
    d1 = np.ascontiguousarray(side1["desc"], np.uint8).reshape(-1, 32)
    d2 = np.ascontiguousarray(side2["desc"], np.uint8).reshape(-1, 32)
    n1, n2 = len(d1), len(d2)
    v1 = None if side1.get("valid") is None else np.ascontiguousarray(side1["valid"], np.uint8)
    v2 = None if side2.get("valid") is None else np.ascontiguousarray(side2["valid"], np.uint8)
    qp, qi, cp, ci = (np.ascontiguousarray(groups[k], np.int32) for k in ("q_ptr", "q_idx", "c_ptr", "c_idx"))
    m12, dd = np.full(max(n1, 1), -1, np.int32), np.full(max(n1, 1), 256, np.int32)
    null_f, null_i, null_u = C.cast(None, _f32p), C.cast(None, _i32p), C.cast(None, _u8p)
    if mode == 1:
        F = np.ascontiguousarray(epi["F12"], np.float32).reshape(9)
        k1 = np.ascontiguousarray(side1["kp_xy"], np.float32).reshape(-1, 2)
        k2 = np.ascontiguousarray(side2["kp_xy"], np.float32).reshape(-1, 2)
        o2 = np.ascontiguousarray(side2["kp_octave"], np.int32)
        sf = np.ascontiguousarray(epi["scale_factor"], np.float32)
        s2 = np.ascontiguousarray(epi["level_sigma2"], np.float32)
        extra = (_ptr(F, _f32p), float(np.float32(epi["ex"])), float(np.float32(epi["ey"])), _ptr(k1, _f32p), _ptr(k2, _f32p),
                 _ptr(o2, _i32p), _ptr(sf, _f32p), _ptr(s2, _f32p))
    else:
        extra = (null_f, 0.0, 0.0, null_f, null_f, null_i, null_f, null_f)
    nm = orb_lib().orb_oracle_bow_search(int(mode), _ptr(d1), n1, _ptr(v1) if v1 is not None else null_u, _ptr(d2), n2,
                                         _ptr(v2) if v2 is not None else null_u, len(qp) - 1, _ptr(qp, _i32p), _ptr(qi, _i32p),
                                         _ptr(cp, _i32p), _ptr(ci, _i32p), int(th), int(bool(th_inclusive)),
                                         float(np.float32(nnratio)), *extra, _ptr(m12, _i32p), _ptr(dd, _i32p))
    return m12[:n1], dd[:n1], nm


def undistort(cam9, xy):
    """cv::undistortPoints(xy, K, D, R = I, P = K) restated (see orb_oracle.cc); cam9 = fx fy cx cy k1 k2 p1 p2 k3."""
    cam9 = np.ascontiguousarray(cam9, np.float32)
    xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
    out = np.zeros_like(xy)
    if len(xy):
        orb_lib().orb_oracle_undistort(_ptr(cam9, _f32p), _ptr(xy, _f32p), len(xy), _ptr(out, _f32p))
    return out


def frame_finish(cam9, kps, min_x, min_y, inv_w, inv_h):
    """Frame::UndistortKeyPoints + AssignFeaturesToGrid: returns (kps_un, cell_start[3073], cell_items)."""
    cam9 = np.ascontiguousarray(cam9, np.float32)
    kps = np.ascontiguousarray(kps)
    n = len(kps)
    un = np.zeros(max(n, 1), KP_DTYPE)
    start, items = np.zeros(64 * 48 + 1, np.int32), np.zeros(max(n, 1), np.int32)
    m = orb_lib().orb_oracle_frame_finish(_ptr(cam9, _f32p), kps.ctypes.data_as(C.c_void_p), n, float(np.float32(min_x)),
                                          float(np.float32(min_y)), float(np.float32(inv_w)), float(np.float32(inv_h)),
                                          un.ctypes.data_as(C.c_void_p), _ptr(start, _i32p), _ptr(items, _i32p))
    return un[:n], start, items[:m]


def matrix(q, t):
    q = np.ascontiguousarray(q, np.uint8).reshape(-1, 32)
    t = np.ascontiguousarray(t, np.uint8).reshape(-1, 32)
    out = np.zeros((len(q), len(t)), np.uint16)
    orb_lib().orb_oracle_matrix(_ptr(q), len(q), _ptr(t), len(t), out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


# ---------------------------------------------------------------------------------------------
# Local bundle adjustment: the restatement (libba_oracle.so) and, where it was built in the
# authoring container, the reference's own g2o (oracle/_ref/libba_ref.so).
# ---------------------------------------------------------------------------------------------
MAX_ITS = 32


class _BaProblem(C.Structure):
    _fields_ = [("n_kf", C.c_int32), ("n_pt", C.c_int32), ("n_edge", C.c_int32),
                ("kf_pose", C.c_void_p), ("kf_fixed", C.c_void_p), ("kf_intr", C.c_void_p),
                ("pt_xyz", C.c_void_p), ("edge_kf", C.c_void_p), ("edge_pt", C.c_void_p),
                ("edge_uv", C.c_void_p), ("edge_inv_sigma2", C.c_void_p), ("edge_ur", C.c_void_p), ("kf_bf", C.c_void_p)]


class _BaOpts(C.Structure):
    _fields_ = [("its_robust", C.c_int32), ("its_final", C.c_int32), ("huber_delta", C.c_double),
                ("chi2_gate", C.c_double), ("stop", C.c_void_p), ("huber_delta_stereo", C.c_double), ("chi2_gate_stereo", C.c_double)]


class _BaStats(C.Structure):
    _fields_ = [("n_its", C.c_int32 * 2), ("chi2", (C.c_double * MAX_ITS) * 2),
                ("lambda_", (C.c_double * MAX_ITS) * 2), ("trials", (C.c_int32 * MAX_ITS) * 2),
                ("chi2_init", C.c_double * 2)]


class _BaResult(C.Structure):
    _fields_ = [("kf_pose", C.c_void_p), ("pt_xyz", C.c_void_p), ("edge_chi2", C.c_void_p),
                ("edge_outlier", C.c_void_p), ("edge_stage1_outlier", C.c_void_p), ("stats", C.c_void_p)]


HUBER_MONO = float(np.float32(np.sqrt(5.991)))  # Optimizer.cc:569: sqrt(5.991) stored in a float

_ba = {}


def ba_ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libba_ref.so"))


def _ba_fn(which):
    if which not in _ba:
        if which == "oracle":
            build()
            L = C.CDLL(os.path.join(HERE, "libba_oracle.so"))
            fn = L.ba_oracle_solve
        else:
            L = C.CDLL(os.path.join(HERE, "_ref", "libba_ref.so"))
            fn = L.ba_ref_solve
        fn.argtypes = [C.POINTER(_BaProblem), C.POINTER(_BaOpts), C.POINTER(_BaResult)]
        _ba[which] = fn
    return _ba[which]


def _ba_call(which, prob, its_robust, its_final, huber_delta, chi2_gate, stop):
    keep = {}
    for k, dt in (("kf_pose", np.float64), ("kf_fixed", np.uint8), ("kf_intr", np.float64), ("pt_xyz", np.float64),
                  ("edge_kf", np.int32), ("edge_pt", np.int32), ("edge_uv", np.float64), ("edge_inv_sigma2", np.float64)):
        keep[k] = np.ascontiguousarray(prob[k], dtype=dt)
    nk, npt, ne = len(keep["kf_fixed"]), len(keep["pt_xyz"]), len(keep["edge_kf"])
    stereo = prob.get("edge_ur") is not None
    if stereo:
        keep["edge_ur"] = np.ascontiguousarray(prob["edge_ur"], dtype=np.float64)
        keep["kf_bf"] = np.ascontiguousarray(prob["kf_bf"], dtype=np.float64)
    p = _BaProblem(nk, npt, ne, *[keep[k].ctypes.data for k in (
        "kf_pose", "kf_fixed", "kf_intr", "pt_xyz", "edge_kf", "edge_pt", "edge_uv", "edge_inv_sigma2")],
        keep["edge_ur"].ctypes.data if stereo else None, keep["kf_bf"].ctypes.data if stereo else None)
    o = _BaOpts(its_robust, its_final, huber_delta, chi2_gate, stop.ctypes.data if stop is not None else None, 0.0, 0.0)
    out = {"kf_pose": np.zeros((nk, 12)), "pt_xyz": np.zeros((npt, 3)), "edge_chi2": np.zeros(ne),
           "edge_outlier": np.zeros(ne, np.uint8), "edge_stage1_outlier": np.zeros(ne, np.uint8)}
    st = _BaStats()
    r = _BaResult(out["kf_pose"].ctypes.data, out["pt_xyz"].ctypes.data, out["edge_chi2"].ctypes.data,
                  out["edge_outlier"].ctypes.data, out["edge_stage1_outlier"].ctypes.data, C.addressof(st))
    rc = _ba_fn(which)(C.byref(p), C.byref(o), C.byref(r))
    assert rc == 0, "BA %s failed" % which
    n = list(st.n_its)
    out["stats"] = {"n_its": n, "chi2": [list(st.chi2[s])[:n[s]] for s in range(2)],
                    "lambda": [list(st.lambda_[s])[:n[s]] for s in range(2)],
                    "trials": [list(st.trials[s])[:n[s]] for s in range(2)], "chi2_init": list(st.chi2_init)}
    return out


def ba_solve(prob, its_robust=5, its_final=10, huber_delta=HUBER_MONO, chi2_gate=5.991, stop=None):
    """The CPU restatement (oracle/ba_oracle.cc)."""
    return _ba_call("oracle", prob, its_robust, its_final, huber_delta, chi2_gate, stop)


def ba_ref_solve(prob, its_robust=5, its_final=10, huber_delta=HUBER_MONO, chi2_gate=5.991, stop=None):
    """The reference's own g2o (authoring container only)."""
    return _ba_call("ref", prob, its_robust, its_final, huber_delta, chi2_gate, stop)


# ---------------------------------------------------------------------------------------------
# Pose-only optimisation (Optimizer::PoseOptimization)
# ---------------------------------------------------------------------------------------------
class _PoseProblem(C.Structure):
    _fields_ = [("n", C.c_int32), ("pose", C.c_void_p), ("intr", C.c_void_p), ("xw", C.c_void_p),
                ("uv", C.c_void_p), ("inv_sigma2", C.c_void_p), ("ur", C.c_void_p), ("bf", C.c_double)]   # stereo: right-image columns (< 0: monocular) and Frame::mbf


class _PoseResult(C.Structure):
    _fields_ = [("pose", C.c_void_p), ("outlier", C.c_void_p), ("n_inliers", C.c_int32),
                ("n_its", C.c_int32 * 4), ("chi2", C.c_double * 4)]


def _pose_call(which, prob):
    if ("pose", which) not in _ba:
        if which == "oracle":
            build()
            fn = C.CDLL(os.path.join(HERE, "libba_oracle.so")).pose_oracle_solve
        else:
            fn = C.CDLL(os.path.join(HERE, "_ref", "libba_ref.so")).pose_ref_solve
        fn.argtypes = [C.POINTER(_PoseProblem), C.POINTER(_PoseResult)]
        _ba[("pose", which)] = fn
    keep = {k: np.ascontiguousarray(prob[k], np.float64) for k in ("pose", "intr", "xw", "uv", "inv_sigma2")}
    n = len(keep["inv_sigma2"])
    if prob.get("ur") is not None:
        keep["ur"] = np.ascontiguousarray(prob["ur"], np.float64)
    p = _PoseProblem(n, *[keep[k].ctypes.data for k in ("pose", "intr", "xw", "uv", "inv_sigma2")],
                     keep["ur"].ctypes.data if "ur" in keep else None, float(prob.get("bf", 0.0)))
    pose = np.zeros(12)
    outl = np.zeros(max(n, 1), np.uint8)
    r = _PoseResult(pose.ctypes.data, outl.ctypes.data, 0)
    assert _ba[("pose", which)](C.byref(p), C.byref(r)) == 0
    return {"pose": pose, "outlier": outl[:n].copy(), "n_inliers": r.n_inliers, "n_its": list(r.n_its), "chi2": list(r.chi2)}


class _Sim3Problem(C.Structure):
    _fields_ = [("n", C.c_int32), ("p1", C.c_void_p), ("p2", C.c_void_p), ("obs1", C.c_void_p), ("obs2", C.c_void_p),
                ("inv_sigma2_1", C.c_void_p), ("inv_sigma2_2", C.c_void_p), ("intr1", C.c_double * 4), ("intr2", C.c_double * 4),
                ("r12", C.c_double * 9), ("t12", C.c_double * 3), ("s12", C.c_double), ("th2", C.c_double), ("fix_scale", C.c_int32)]


class _Sim3Result(C.Structure):
    _fields_ = [("r12", C.c_double * 9), ("t12", C.c_double * 3), ("s12", C.c_double), ("inlier", C.c_void_p), ("n_inliers", C.c_int32),
                ("n_its", C.c_int32 * 2), ("chi2", C.c_double * 2)]


def sim3_problem_struct(prob, cls=_Sim3Problem):
    """ctypes view of a synth.synth_sim3-style dict (shared with api.py through the same field layout)."""
    keep = {k: np.ascontiguousarray(prob[k], np.float64) for k in ("p1", "p2", "obs1", "obs2", "inv_sigma2_1", "inv_sigma2_2")}
    p = cls()
    p.n = len(keep["inv_sigma2_1"])
    for k, a in keep.items():
        setattr(p, k, a.ctypes.data)
    p.intr1 = (C.c_double * 4)(*[float(v) for v in prob["intr1"]])
    p.intr2 = (C.c_double * 4)(*[float(v) for v in prob["intr2"]])
    p.r12 = (C.c_double * 9)(*[float(v) for v in np.asarray(prob["r12"]).reshape(9)])
    p.t12 = (C.c_double * 3)(*[float(v) for v in prob["t12"]])
    p.s12, p.th2, p.fix_scale = float(prob["s12"]), float(prob["th2"]), int(prob["fix_scale"])
    return p, keep


def _sim3_call(which, prob):
    if ("sim3", which) not in _ba:
        build()
        if which == "oracle":
            fn = C.CDLL(os.path.join(HERE, "libba_oracle.so")).sim3_oracle_solve
        else:
            fn = C.CDLL(os.path.join(HERE, "_ref", "libba_ref.so")).sim3_ref_solve
        fn.argtypes = [C.POINTER(_Sim3Problem), C.POINTER(_Sim3Result)]
        _ba[("sim3", which)] = fn
    p, keep = sim3_problem_struct(prob)
    inl = np.zeros(max(p.n, 1), np.uint8)
    r = _Sim3Result()
    r.inlier = inl.ctypes.data
    assert _ba[("sim3", which)](C.byref(p), C.byref(r)) == 0
    del keep
    return {"r12": np.array(r.r12[:]).reshape(3, 3), "t12": np.array(r.t12[:]), "s12": r.s12, "inlier": inl[:p.n].copy(),
            "n_inliers": r.n_inliers, "n_its": list(r.n_its), "chi2": list(r.chi2)}


def sim3_solve(prob):
    """CPU oracle restatement of Optimizer::OptimizeSim3 (oracle/ba_oracle.cc)."""
    return _sim3_call("oracle", prob)


def sim3_ref_solve(prob):
    """The reference's own g2o (oracle/_ref/libba_ref.so).  Authoring container only."""
    return _sim3_call("ref", prob)


def pose_solve(prob):
    """CPU restatement of PoseOptimization (oracle/ba_oracle.cc)."""
    return _pose_call("oracle", prob)


def pose_ref_solve(prob):
    """The reference's own g2o (authoring container only)."""
    return _pose_call("ref", prob)
