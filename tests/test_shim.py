"""The C++ class-surface shim (weiner_slamit_v2_amd/shim): ORBextractor / ORBmatcher / Optimizer
with the reference's names and signatures, driven like the reference's callers drive them.

CPU part: the shim builds against libslamit_hip.so with plain g++.  GPU part: shim_test runs the
three surfaces and the results are compared with the oracle.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

from tests.helpers import ROOT, load_ba_golden, load_pose_golden

SHIM = os.path.join(ROOT, "weiner_slamit_v2_amd", "shim")
EXE = os.path.join(SHIM, "shim_test")


def _build():
    from weiner_slamit_v2_amd import build as hb

    hb.build()
    subprocess.check_call(["make", "-s", "-C", SHIM, "-f", "Makefile", "all"])


def test_shim_builds_and_keeps_the_reference_surface():
    _build()
    assert os.path.exists(EXE) and os.path.exists(os.path.join(SHIM, "libslamit_shim.so"))
    syms = subprocess.check_output(["nm", "-DC", os.path.join(SHIM, "libslamit_shim.so")]).decode()
    for want in ("ORB_SLAM2::ORBextractor::ORBextractor(int, float, int, int, int)",
                 "ORB_SLAM2::ORBextractor::operator()(",
                 "ORB_SLAM2::ORBmatcher::ORBmatcher(float, bool)",
                 "ORB_SLAM2::ORBmatcher::DescriptorDistance(cv::Mat const&, cv::Mat const&)",
                 "ORB_SLAM2::ORBmatcher::TH_LOW", "ORB_SLAM2::ORBmatcher::TH_HIGH", "ORB_SLAM2::ORBmatcher::HISTO_LENGTH"):
        assert want in syms, want
    hdr = open(os.path.join(SHIM, "Optimizer.h")).read()
    assert "static void LocalBundleAdjustment(KeyFrameT* pKF, bool* pbStopFlag, MapT* pMap)" in hdr


@pytest.mark.gpu
def test_shim_orbextractor(tmp_path):
    from oracle import bindings as ob
    from weiner_slamit_v2_amd import synth

    _build()
    img = synth.synth_frame(640, 480, 77)
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    img.tofile(raw)
    subprocess.check_call([EXE, "orb", str(raw), "640", "480", str(out)])
    buf = open(out, "rb").read()
    n = struct.unpack_from("<i", buf, 0)[0]
    kps = np.frombuffer(buf, ob.KP_DTYPE, n, 4)
    desc = np.frombuffer(buf, np.uint8, n * 32, 4 + 28 * n).reshape(n, 32)
    orc = ob.OrbOracle(1000)
    ko, do = orc.extract(img)
    assert n == len(ko) and np.array_equal(desc, do)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        assert np.array_equal(kps[f], ko[f])
    assert np.array_equal(kps["angle"].view(np.uint32), ko["angle"].view(np.uint32))
    off = 4 + 60 * n
    nl = struct.unpack_from("<i", buf, off)[0]
    off += 4
    assert nl == 8
    for l in range(nl):
        w, h, s = struct.unpack_from("<iiI", buf, off)
        off += 12
        assert (w, h) == orc.level_size(l)
        plane = orc.level(l)  # padded (h+38, w+38): the shim's ROI must sit inside an identical frame
        want = 0
        for v in plane.reshape(-1).astype(np.uint64):
            want = (want * 31 + int(v)) & 0xFFFFFFFF
        assert s == want, "mvImagePyramid[%d] (ROI + REFLECT_101 frame) differs" % l
    sf = np.frombuffer(buf, np.float32, nl, off)
    assert np.array_equal(sf, orc.tables()["scale"])
    assert np.array_equal(np.frombuffer(buf, np.float32, nl, off + 4 * nl), orc.tables()["inv_sigma2"])


@pytest.mark.gpu
def test_shim_orbmatcher(tmp_path):
    from oracle import bindings as ob

    _build()
    rs = np.random.RandomState(3)
    a = rs.randint(0, 256, (300, 32)).astype(np.uint8)
    b = rs.randint(0, 256, (280, 32)).astype(np.uint8)
    b[:100] = a[100:200] ^ (rs.randint(0, 256, (100, 32)) < 8).astype(np.uint8)  # near duplicates
    pa, pb, out = tmp_path / "a.bin", tmp_path / "b.bin", tmp_path / "o.bin"
    a.tofile(pa)
    b.tofile(pb)
    subprocess.check_call([EXE, "match", str(pa), "300", str(pb), "280", str(out)])
    r = np.fromfile(out, np.int32)
    idx, best, second, m12 = r[:300], r[300:600], r[600:900], r[900:1200]
    oi, obest, osec = ob.best2(a, b)
    assert np.array_equal(idx, oi) and np.array_equal(best, obest) and np.array_equal(second, osec)
    ok = (obest <= 50) & (obest.astype(np.float32) < np.float32(0.9) * osec.astype(np.float32))
    assert np.array_equal(m12, np.where(ok, oi, -1)) and r[1200] == ok.sum() and ok.sum() >= 90
    assert r[1201] == ob.distance(a[0], b[0])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["fixed3", "stereo_mixed"])
def test_shim_optimizer_local_ba(tmp_path, name):
    """LocalMapping.cc:84's call against mock KeyFrame / MapPoint / Map types; "stereo_mixed": keyframes with mvuRight >= 0 on half
    of their keypoints and an mbf (the EdgeStereoSE3ProjectXYZ edges of Optimizer.cc:621-650), against the reference-g2o golden."""
    _build()
    prob, ref = load_ba_golden(os.path.join(ROOT, "tests", "golden", "ba_%s.npz" % name))
    K, P, E = len(prob["kf_fixed"]), len(prob["pt_xyz"]), len(prob["edge_kf"])
    blob = struct.pack("<iii", K, P, E)
    blob += prob["kf_pose"].astype(np.float32).tobytes()
    blob += prob["kf_fixed"].tobytes() + b"\0" * ((4 - K % 4) % 4)
    blob += prob["kf_intr"][0].astype(np.float32).tobytes()
    blob += prob["pt_xyz"].astype(np.float32).tobytes()
    blob += prob["edge_kf"].astype(np.int32).tobytes() + prob["edge_pt"].astype(np.int32).tobytes()
    blob += prob["edge_uv"].astype(np.float32).tobytes() + prob["edge_inv_sigma2"].astype(np.float32).tobytes()
    if "edge_ur" in prob:
        blob += prob["kf_bf"][:1].astype(np.float32).tobytes() + prob["edge_ur"].astype(np.float32).tobytes()
    pin, pout = tmp_path / "p.bin", tmp_path / "o.bin"
    open(pin, "wb").write(blob)
    subprocess.check_call([EXE, "ba", str(pin), str(pout)])
    raw = open(pout, "rb").read()
    f = np.frombuffer(raw, np.float32, 12 * K + 3 * P)
    R = f[:9 * K].reshape(K, 9)
    t = f[9 * K:12 * K].reshape(K, 3)
    pts = f[12 * K:].reshape(P, 3)
    erased, updates = struct.unpack_from("<ii", raw, 4 * (12 * K + 3 * P))
    # the reference writes float32 poses / points back (Optimizer.cc:762-777): compare at that precision
    assert np.abs(R - ref["kf_pose"][:, :9]).max() < 2e-6
    assert np.abs(t - ref["kf_pose"][:, 9:]).max() < 1e-5 * max(np.abs(ref["kf_pose"][:, 9:]).max(), 1)
    assert np.abs(pts - ref["pt_xyz"]).max() < 1e-5 * np.abs(ref["pt_xyz"]).max()
    assert erased == int(ref["edge_outlier"].sum()) and updates == P


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["typical", "stereo_mixed"])
def test_shim_optimizer_pose_optimization(tmp_path, name):
    """Tracking's Optimizer::PoseOptimization(&mCurrentFrame) against a mock Frame; "stereo_mixed": half of the keypoints carry
    mvuRight >= 0 and the frame an mbf (EdgeStereoSE3ProjectXYZOnlyPose edges, Optimizer.cc:319-356), reference-g2o golden."""
    _build()
    prob, ref = load_pose_golden(os.path.join(ROOT, "tests", "golden", "pose_%s.npz" % name))
    n = len(prob["inv_sigma2"])
    blob = struct.pack("<i", n) + prob["pose"].astype(np.float32).tobytes() + prob["intr"].astype(np.float32).tobytes()
    blob += prob["xw"].astype(np.float32).tobytes() + prob["uv"].astype(np.float32).tobytes() + prob["inv_sigma2"].astype(np.float32).tobytes()
    if "ur" in prob:
        blob += struct.pack("<f", prob["bf"]) + prob["ur"].astype(np.float32).tobytes()
    pin, pout = tmp_path / "p.bin", tmp_path / "o.bin"
    open(pin, "wb").write(blob)
    subprocess.check_call([EXE, "pose", str(pin), str(pout)])
    raw = open(pout, "rb").read()
    inl = struct.unpack_from("<i", raw, 0)[0]
    T = np.frombuffer(raw, np.float32, 12, 4).reshape(3, 4)
    outl = np.frombuffer(raw, np.uint8, n, 52)
    assert inl == ref["n_inliers"] and np.array_equal(outl, ref["outlier"])
    want = np.concatenate([ref["pose"][:9].reshape(3, 3), ref["pose"][9:].reshape(3, 1)], 1)
    assert np.abs(T - want).max() < 1e-5 * max(np.abs(want).max(), 1.0)  # float32 write-back like the reference


def _search_frame_blob(variant, frame, n, m, th, nnratio, scale, angle, state, max_x, max_y):
    blob = struct.pack("<iiiff", variant, n, m, th, nnratio)
    blob += struct.pack("<6f", frame["min_x"], max_x, frame["min_y"], max_y, frame["inv_w"], frame["inv_h"])
    blob += scale.tobytes() + frame["kp_xy"].astype(np.float32).tobytes() + frame["kp_octave"].astype(np.int32).tobytes()
    blob += angle.astype(np.float32).tobytes() + state.astype(np.int32).tobytes() + frame["desc"].tobytes()
    return blob


def _apply_in_order(n, state, match):
    """What the reference's loop leaves in mvpMapPoints: a later query overwrites an earlier one (possible only
    when the earlier map point had no observations)."""
    owner = np.where(state > 0, -2, -1).astype(np.int32)
    for q, k in enumerate(match):
        if k >= 0:
            owner[k] = q
    return owner


def _gemm_row(Rrow, t, X, Y, Z):
    """One row of cv::Mat x3Dc = Rcw * x3Dw + tcw on CV_32F operands as OpenCV 2.4's cv::gemm evaluates a 3 x 3 by 3 x 1 product
    without flags (matmul.cpp's small-matrix branch): the dot product in float32, left to right, then one double add of t and a
    rounding to float32 (ORBmatcher.cc:1363, 851, 1497).  Restated here in numpy, independently of the shim's slamit_gemm_row3;
    parity unpinned against OpenCV itself (its source is not in the reference tree)."""
    f32, f64 = np.float32, np.float64
    t0 = (f32(Rrow[0]) * X.astype(f32) + f32(Rrow[1]) * Y.astype(f32)).astype(f32) + f32(Rrow[2]) * Z.astype(f32)
    return (t0.astype(f32).astype(f64) + f64(f32(t))).astype(f32)


@pytest.mark.gpu
def test_shim_search_by_projection_local_map(tmp_path):
    """ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th) through the template + mock types."""
    from oracle import bindings as ob
    from weiner_slamit_v2_amd import synth

    _build()
    n, m, th, nnratio = 1800, 700, 3.0, 0.8
    frame, qs = synth.synth_search(n, m, 31)
    rs = np.random.RandomState(5)
    scale = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    state = np.where(frame["kp_taken"] > 0, 1, rs.randint(0, 2, n) * 2).astype(np.int32)   # 1 taken, 2 = point w/o observations
    viewcos = rs.choice(np.array([0.9995, 0.95], np.float32), m)
    level = rs.randint(0, 8, m).astype(np.int32)
    inview = (rs.rand(m) < 0.9).astype(np.int32)
    bad = (rs.rand(m) < 0.05).astype(np.int32)
    nobs = (rs.rand(m) < 0.95).astype(np.int32) * 3
    proj = qs["uvr"][:, :2].astype(np.float32)
    blob = _search_frame_blob(0, frame, n, m, th, nnratio, scale, np.zeros(n, np.float32), state, 645.1, 483.9)
    blob += proj.tobytes() + viewcos.tobytes() + level.tobytes() + inview.tobytes() + bad.tobytes() + nobs.tobytes()
    blob += qs["desc"].tobytes()
    pin, pout = tmp_path / "s.bin", tmp_path / "o.bin"
    open(pin, "wb").write(blob)
    subprocess.check_call([EXE, "search", str(pin), str(pout)])
    r = np.fromfile(pout, np.int32)
    assert r[0] == 0
    # expected: the same queries, in order, through the oracle
    keep = np.nonzero((inview != 0) & (bad == 0))[0]
    rad = (np.where(viewcos > np.float32(0.998), np.float32(2.5), np.float32(4.0)).astype(np.float32) * np.float32(th)) * scale[level]
    q = dict(uvr=np.concatenate([proj, rad[:, None]], 1)[keep], level_min=level[keep] - 1, level_max=level[keep],
             desc=qs["desc"][keep], takes=(nobs[keep] > 0).astype(np.uint8))
    f = dict(frame, kp_taken=(state == 1).astype(np.uint8))
    match, nm, _ = ob.guided_search(f, q, 100, True, nnratio)
    assert r[1] == nm and nm > 50
    full = np.full(m, -1, np.int32)
    full[keep] = match
    assert np.array_equal(r[2:], _apply_in_order(n, state, full))


@pytest.mark.gpu
def test_shim_search_by_projection_last_frame(tmp_path):
    """ORBmatcher::SearchByProjection(CurrentFrame, LastFrame, th, bMono): projection on the host, search on the
    device, rotation histogram on the host."""
    from oracle import bindings as ob
    from weiner_slamit_v2_amd import synth

    _build()
    f32 = np.float32
    n, m, th = 1500, 1200, 15.0
    rs = np.random.RandomState(8)
    frame, _ = synth.synth_search(n, 4, 32)
    scale = (f32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    fx, fy, cx, cy = f32(520.9), f32(521.0), f32(325.1), f32(249.7)
    Rc, tc = synth.se3_exp(np.array([0.01, -0.02, 0.015, 0.03, -0.01, 0.02]))
    Rc, tc = Rc.astype(np.float32), tc.astype(np.float32)
    # last-frame map points: back-project current keypoints (with jitter) to depth 2..8 in the CURRENT camera
    src = rs.randint(0, n, m)
    depth = rs.uniform(2, 8, m)
    px = frame["kp_xy"][src].astype(np.float64) + rs.uniform(-6, 6, (m, 2))
    pc = np.stack([(px[:, 0] - cx) / fx * depth, (px[:, 1] - cy) / fy * depth, depth], 1)
    pc[::50] *= -1                                      # some behind the camera
    world = ((pc - tc.astype(np.float64)) @ Rc.astype(np.float64)).astype(np.float32)   # R^T (pc - t)
    loct = frame["kp_octave"][src].astype(np.int32)
    angle_cur = rs.uniform(0, 360, n).astype(np.float32)
    lang = ((angle_cur[src] + np.where(rs.rand(m) < 0.8, 12.0, rs.uniform(0, 360, m))) % 360.0).astype(np.float32)
    has = (rs.rand(m) < 0.9).astype(np.int32)
    outl = (rs.rand(m) < 0.05).astype(np.int32)
    nobs = (rs.rand(m) < 0.95).astype(np.int32) * 2
    qdesc = frame["desc"][src].copy()
    flip = rs.randint(0, 256, (m, 40))
    for j in range(m):
        for b in flip[j, :rs.randint(0, 40)]:
            qdesc[j, b >> 3] ^= np.uint8(1 << (b & 7))
    state = np.where(frame["kp_taken"] > 0, 1, 0).astype(np.int32)
    Tcw = np.concatenate([Rc.reshape(-1), tc]).astype(np.float32)
    blob = _search_frame_blob(1, frame, n, m, th, 0.9, scale, angle_cur, state, 645.1, 483.9)
    blob += Tcw.tobytes() + Tcw.tobytes() + np.array([fx, fy, cx, cy, 0.08], np.float32).tobytes() + struct.pack("<i", 1)
    blob += world.tobytes() + lang.tobytes() + has.tobytes() + outl.tobytes() + loct.tobytes() + nobs.tobytes() + qdesc.tobytes()
    pin, pout = tmp_path / "s.bin", tmp_path / "o.bin"
    open(pin, "wb").write(blob)
    subprocess.check_call([EXE, "search", str(pin), str(pout)])
    r = np.fromfile(pout, np.int32)
    assert r[0] == 0
    # expected, float32 step by step as ORBmatcher.cc:1363-1386 computes it
    X, Y, Z = world[:, 0], world[:, 1], world[:, 2]
    xc = _gemm_row(Rc[0], tc[0], X, Y, Z)
    yc = _gemm_row(Rc[1], tc[1], X, Y, Z)
    zc = _gemm_row(Rc[2], tc[2], X, Y, Z)
    with np.errstate(divide="ignore"):
        invz = (1.0 / zc.astype(np.float64)).astype(np.float32)
    u = (fx * xc) * invz + cx
    v = (fy * yc) * invz + cy
    ok = (has != 0) & (outl == 0) & ~(invz < 0) & ~(u < f32(frame["min_x"])) & ~(u > f32(645.1)) & ~(v < f32(frame["min_y"])) & ~(v > f32(483.9))
    keep = np.nonzero(ok)[0]
    q = dict(uvr=np.stack([u, v, f32(th) * scale[loct]], 1)[keep], level_min=loct[keep] - 1, level_max=loct[keep] + 1,
             desc=qdesc[keep], takes=(nobs[keep] > 0).astype(np.uint8))
    match, nm, _ = ob.guided_search(dict(frame, kp_taken=(state == 1).astype(np.uint8)), q, 100, False, 0.9)
    full = np.full(m, -1, np.int32)
    full[keep] = match
    owner = _apply_in_order(n, state, full)
    # rotation histogram (ORBmatcher.cc:1436-1469): bins of bestIdx2 in visiting order
    hist = [[] for _ in range(30)]
    factor = f32(1.0) / f32(30)
    for qi in keep:
        k = full[qi]
        if k < 0:
            continue
        rot = f32(lang[qi] - angle_cur[k])
        if rot < 0:
            rot = f32(rot + f32(360))
        b = int(np.floor(f32(rot * factor) + f32(0.5)))
        hist[0 if b == 30 else b].append(k)
    sizes = [len(h) for h in hist]
    order = sorted(range(30), key=lambda i: -sizes[i])
    top = max(sizes)
    assert top > 100 and sizes[order[1]] < 0.1 * top, "test premise: one dominant rotation bin"
    for i in range(30):
        if i != order[0]:
            for k in hist[i]:
                owner[k] = -1
                nm -= 1
    assert r[1] == nm
    assert np.array_equal(r[2:], owner)


@pytest.mark.gpu
def test_shim_frame_ops(tmp_path):
    """FrameOps::ComputeImageBounds + UndistortAndAssign on a mock Frame == the oracle's Frame epilogue."""
    from oracle import bindings as ob

    _build()
    cam = [526.69, 540.36, 313.07, 238.39, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
    rs = np.random.RandomState(12)
    n = 900
    kps = np.zeros(n, ob.KP_DTYPE)
    kps["x"], kps["y"] = rs.uniform(16, 624, n).astype(np.float32), rs.uniform(16, 464, n).astype(np.float32)
    kps["octave"], kps["angle"], kps["size"] = rs.randint(0, 8, n), rs.uniform(0, 360, n), 31.0
    kps["x"][:2], kps["y"][:2] = [640, 0], [480, 0]
    pin, pout = tmp_path / "f.bin", tmp_path / "o.bin"
    open(pin, "wb").write(struct.pack("<iii", n, 640, 480) + np.asarray(cam, np.float32).tobytes() + kps.tobytes())
    subprocess.check_call([EXE, "frame", str(pin), str(pout)])
    raw = open(pout, "rb").read()
    assert struct.unpack_from("<i", raw, 0)[0] == 0
    b = np.frombuffer(raw, np.float32, 6, 4)
    c = ob.undistort(cam, [[0, 0], [640, 0], [0, 480], [640, 480]])
    want = [min(c[0, 0], c[2, 0]), max(c[1, 0], c[3, 0]), min(c[0, 1], c[1, 1]), max(c[2, 1], c[3, 1])]
    assert np.array_equal(b[:4], np.asarray(want, np.float32))
    assert b[4] == np.float32(64) / np.float32(b[1] - b[0]) and b[5] == np.float32(48) / np.float32(b[3] - b[2])
    un = np.frombuffer(raw, ob.KP_DTYPE, n, 28)
    ou, os_, oi = ob.frame_finish(cam, kps, b[0], b[2], b[4], b[5])
    assert np.array_equal(un.view(np.uint8), ou.view(np.uint8))
    counts = np.frombuffer(raw, np.int32, 64 * 48, 28 + 28 * n)
    assert np.array_equal(counts, np.diff(os_))
    items = np.frombuffer(raw, np.int32, int(counts.sum()), 28 + 28 * n + 4 * 64 * 48)
    assert np.array_equal(items, oi)


@pytest.mark.gpu
def test_shim_fuse(tmp_path):
    """ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>, th) through the template + mock types: projection and checks on the
    host, the gated window search on the device, Replace / AddObservation bookkeeping in order."""
    from oracle import bindings as ob
    from weiner_slamit_v2_amd import synth

    _build()
    f32 = np.float32
    rs = np.random.RandomState(21)
    n, m, th = 1400, 900, 3.0
    frame, _ = synth.synth_search(n, 4, 33)
    scale = (f32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    invsig = (f32(1) / (scale * scale)).astype(np.float32)
    fx, fy, cx, cy = f32(526.69), f32(540.36), f32(313.07), f32(238.39)
    Rc, tc = synth.se3_exp(np.array([0.02, -0.01, 0.03, 0.1, -0.05, 0.02]))
    Rc, tc = Rc.astype(np.float32), tc.astype(np.float32)
    Ow = (-(Rc.astype(np.float64).T @ tc.astype(np.float64))).astype(np.float32)
    src = rs.randint(0, n, m)
    depth = rs.uniform(2, 8, m)
    px = frame["kp_xy"][src].astype(np.float64) + rs.normal(0, 1.2, (m, 2))
    pc = np.stack([(px[:, 0] - cx) / fx * depth, (px[:, 1] - cy) / fy * depth, depth], 1)
    pc[::40] *= -1                                   # behind the camera
    pos = ((pc - tc.astype(np.float64)) @ Rc.astype(np.float64)).astype(np.float32)
    PO = pos.astype(np.float32) - Ow
    d3 = np.sqrt((PO.astype(np.float64) ** 2).sum(1)).astype(np.float32)
    normal = (PO / np.maximum(d3[:, None], 1e-6)).astype(np.float32)
    normal[::17] *= -1                               # viewing angle check fails
    maxd, mind = (d3 * f32(1.5)).astype(np.float32), (d3 * f32(0.6)).astype(np.float32)
    mind[::23] = d3[::23] * f32(1.2)                 # outside the scale-invariance range
    level = np.clip(frame["kp_octave"][src] + rs.randint(0, 2, m), 0, 7).astype(np.int32)
    nobs = rs.randint(1, 6, m).astype(np.int32)
    bad = (rs.rand(m) < 0.05).astype(np.int32)
    inkf = (rs.rand(m) < 0.05).astype(np.int32)
    isnull = (rs.rand(m) < 0.03).astype(np.int32)
    kf_state = np.where(rs.rand(n) < 0.4, rs.randint(1, 7, n), 0).astype(np.int32)   # k-1 observations of the KF's own point
    qdesc = frame["desc"][src].copy()
    flip = rs.randint(0, 256, (m, 30))
    for j in range(m):
        for b in flip[j, :rs.randint(0, 30)]:
            qdesc[j, b >> 3] ^= np.uint8(1 << (b & 7))
    bounds = np.array([frame["min_x"], 645.1, frame["min_y"], 483.9, frame["inv_w"], frame["inv_h"]], np.float32)
    blob = struct.pack("<iif", n, m, th) + Rc.tobytes() + tc.tobytes() + Ow.tobytes() + np.array([fx, fy, cx, cy], np.float32).tobytes()
    blob += bounds.tobytes() + scale.tobytes() + invsig.tobytes()
    blob += frame["kp_xy"].astype(np.float32).tobytes() + frame["kp_octave"].astype(np.int32).tobytes() + kf_state.tobytes() + frame["desc"].tobytes()
    blob += pos.tobytes() + normal.tobytes() + maxd.tobytes() + mind.tobytes() + level.tobytes() + nobs.tobytes() + bad.tobytes()
    blob += inkf.tobytes() + isnull.tobytes() + qdesc.tobytes()
    pin, pout = tmp_path / "f.bin", tmp_path / "o.bin"
    open(pin, "wb").write(blob)
    subprocess.check_call([EXE, "fuse", str(pin), str(pout)])
    r = np.fromfile(pout, np.int32)
    assert r[0] == 0
    # expected: the reference's float arithmetic step by step (ORBmatcher.cc:851-895)
    X, Y, Z = pos[:, 0], pos[:, 1], pos[:, 2]
    xc = _gemm_row(Rc[0], tc[0], X, Y, Z)
    yc = _gemm_row(Rc[1], tc[1], X, Y, Z)
    zc = _gemm_row(Rc[2], tc[2], X, Y, Z)
    with np.errstate(divide="ignore", invalid="ignore"):
        invz = f32(1) / zc
    u, v = fx * (xc * invz) + cx, fy * (yc * invz) + cy
    PO = np.stack([X - Ow[0], Y - Ow[1], Z - Ow[2]], 1)
    dist3D = np.sqrt((PO.astype(np.float64) ** 2).sum(1)).astype(np.float32)
    dot = (PO.astype(np.float64) * normal.astype(np.float64)).sum(1)
    ok = (isnull == 0) & (bad == 0) & (inkf == 0) & ~(zc < 0) & (u >= bounds[0]) & (u < bounds[1]) & (v >= bounds[2]) & (v < bounds[3])
    ok &= ~((dist3D < mind) | (dist3D > maxd)) & ~(dot < 0.5 * dist3D.astype(np.float64))
    keep = np.nonzero(ok)[0]
    assert 200 < len(keep) < m
    q = dict(uvr=np.stack([u, v, f32(th) * scale[level]], 1)[keep], level_min=level[keep] - 1, level_max=level[keep], desc=qdesc[keep],
             takes=np.zeros(len(keep), np.uint8))
    match, _, _ = ob.guided_search(dict(frame, kp_taken=np.zeros(n, np.uint8)), q, 50, False, 0.6, 5.99, invsig)
    # the bookkeeping of ORBmatcher.cc:955-976, in order
    owner = np.where(kf_state > 0, -2, -1).astype(np.int64)
    own_obs = kf_state - 1
    own_bad = np.zeros(n, bool)
    p_bad, p_inkf, p_nobs = bad.astype(bool).copy(), inkf.astype(bool).copy(), nobs.copy()
    added, replaced, own_replaced = np.full(m, -1), np.full(m, -1), np.full(n, -1)
    nfused = 0
    for k, j in enumerate(keep):
        bi = match[k]
        if bi < 0 or p_bad[j] or p_inkf[j]:
            continue
        if owner[bi] == -2:                      # the keyframe's own point sits there
            if not own_bad[bi]:
                if own_obs[bi] > p_nobs[j]:
                    replaced[j], p_bad[j] = -2, True
                else:
                    own_replaced[bi], own_bad[bi] = j, True
        elif owner[bi] >= 0:                     # a map point added earlier in this call
            o = owner[bi]
            if not p_bad[o]:
                if p_nobs[o] > p_nobs[j]:
                    replaced[j], p_bad[j] = o, True
                else:
                    replaced[o], p_bad[o] = j, True
        else:
            added[j], p_inkf[j], owner[bi] = bi, True, j
            p_nobs[j] += 1
        nfused += 1
    assert r[1] == nfused and nfused > 100
    per = r[2:2 + 3 * m].reshape(m, 3)
    assert np.array_equal(per[:, 0], added) and np.array_equal(per[:, 1], replaced) and np.array_equal(per[:, 2], p_bad.astype(np.int32))
    assert np.array_equal(r[2 + 3 * m:2 + 3 * m + n], owner)
    assert np.array_equal(r[2 + 3 * m + n:], own_replaced)


@pytest.mark.gpu
def test_shim_search_for_initialization(tmp_path):
    """ORBmatcher::SearchForInitialization through the template: device matching loop (mode 1), host rotation histogram
    over the matches accepted at their own turn, vbPrevMatched update."""
    from oracle import bindings as ob
    from weiner_slamit_v2_amd import synth

    _build()
    f32 = np.float32
    f1, prev, f2 = synth.synth_init_pair(1200, 7)
    n1, n2, window, nnratio = len(f1["kp_octave"]), len(f2["kp_octave"]), 40, 0.9
    m12, nm, acc = ob.search_for_initialization(f1, prev, f2, window, nnratio, 50)
    # angles: one dominant rotation (12 degrees) for 80 % of the accepted matches
    rs = np.random.RandomState(3)
    a2 = f2["angle"].astype(np.float32)
    a1 = f1["angle"].astype(np.float32).copy()
    hit = np.nonzero(acc >= 0)[0]
    a1[hit] = ((a2[acc[hit]] + np.where(rs.rand(len(hit)) < 0.8, 12.0, rs.uniform(0, 360, len(hit)))) % 360.0).astype(np.float32)
    bounds = np.array([f2["min_x"], 645.1, f2["min_y"], 483.9, f2["inv_w"], f2["inv_h"]], np.float32)
    blob = struct.pack("<iiif", n1, n2, window, nnratio) + bounds.tobytes()
    blob += f1["kp_octave"].astype(np.int32).tobytes() + a1.tobytes() + prev.astype(np.float32).tobytes() + f1["desc"].tobytes()
    blob += f2["kp_xy"].astype(np.float32).tobytes() + f2["kp_octave"].astype(np.int32).tobytes() + a2.tobytes() + f2["desc"].tobytes()
    pin, pout = tmp_path / "i.bin", tmp_path / "o.bin"
    open(pin, "wb").write(blob)
    subprocess.check_call([EXE, "init", str(pin), str(pout)])
    raw = open(pout, "rb").read()
    status, got_nm = struct.unpack_from("<ii", raw, 0)
    got12 = np.frombuffer(raw, np.int32, n1, 8)
    got_prev = np.frombuffer(raw, np.float32, 2 * n1, 8 + 4 * n1).reshape(n1, 2)
    assert status == 0
    # expected: histogram over `acc` in i1 order (ORBmatcher.cc:467-477), three maxima, prune, update prev
    hist = [[] for _ in range(30)]
    factor = f32(1.0) / f32(30)
    for i1 in hit:
        rot = f32(a1[i1] - a2[acc[i1]])
        if rot < 0:
            rot = f32(rot + f32(360))
        b = int(np.floor(f32(rot * factor) + f32(0.5)))
        hist[0 if b == 30 else b].append(i1)
    sizes = [len(h) for h in hist]
    order = sorted(range(30), key=lambda i: -sizes[i])
    assert sizes[order[0]] > 100 and sizes[order[1]] < 0.1 * sizes[order[0]], "test premise: one dominant rotation bin"
    e12, enm = m12.copy(), nm
    for i in range(30):
        if i != order[0]:
            for i1 in hist[i]:
                if e12[i1] >= 0:
                    e12[i1] = -1
                    enm -= 1
    assert got_nm == enm and np.array_equal(got12, e12)
    eprev = prev.astype(np.float32).copy()
    ok = e12 >= 0
    eprev[ok] = f2["kp_xy"][e12[ok]]
    assert np.array_equal(got_prev, eprev)


@pytest.mark.gpu
def test_shim_search_by_projection_relocalization(tmp_path):
    """ORBmatcher::SearchByProjection(CurrentFrame, KeyFrame*, sAlreadyFound, th, ORBdist) (Tracking::Relocalization)."""
    from oracle import bindings as ob
    from weiner_slamit_v2_amd import synth

    _build()
    f32 = np.float32
    n, m, th, orbdist = 1500, 1000, 10.0, 100
    rs = np.random.RandomState(18)
    frame, _ = synth.synth_search(n, 4, 34)
    scale = (f32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
    fx, fy, cx, cy = f32(520.9), f32(521.0), f32(325.1), f32(249.7)
    Rc, tc = synth.se3_exp(np.array([-0.02, 0.015, 0.01, 0.05, 0.02, -0.03]))
    Rc, tc = Rc.astype(np.float32), tc.astype(np.float32)
    Ow = (-(Rc.astype(np.float64).T @ tc.astype(np.float64))).astype(np.float32)   # -Rcw.t() * tcw: double sum, one rounding
    src = rs.randint(0, n, m)
    depth = rs.uniform(2, 8, m)
    px = frame["kp_xy"][src].astype(np.float64) + rs.uniform(-5, 5, (m, 2))
    pc = np.stack([(px[:, 0] - cx) / fx * depth, (px[:, 1] - cy) / fy * depth, depth], 1)
    world = ((pc - tc.astype(np.float64)) @ Rc.astype(np.float64)).astype(np.float32)
    level = np.clip(frame["kp_octave"][src] + rs.randint(-1, 2, m), 0, 7).astype(np.int32)
    angle_cur = rs.uniform(0, 360, n).astype(np.float32)
    kang = ((angle_cur[src] + np.where(rs.rand(m) < 0.8, 12.0, rs.uniform(0, 360, m))) % 360.0).astype(np.float32)
    has, bad, found = (rs.rand(m) < 0.9).astype(np.int32), (rs.rand(m) < 0.05).astype(np.int32), (rs.rand(m) < 0.1).astype(np.int32)
    X, Y, Z = world[:, 0], world[:, 1], world[:, 2]
    PO = np.stack([X - Ow[0], Y - Ow[1], Z - Ow[2]], 1)
    d3 = np.sqrt((PO.astype(np.float64) ** 2).sum(1)).astype(np.float32)
    maxd, mind = (d3 * f32(1.4)).astype(np.float32), (d3 * f32(0.7)).astype(np.float32)
    maxd[::19] = d3[::19] * f32(0.9)
    qdesc = frame["desc"][src].copy()
    flip = rs.randint(0, 256, (m, 40))
    for j in range(m):
        for b in flip[j, :rs.randint(0, 40)]:
            qdesc[j, b >> 3] ^= np.uint8(1 << (b & 7))
    state = np.where(frame["kp_taken"] > 0, 1, np.where(rs.rand(n) < 0.1, 2, 0)).astype(np.int32)
    Tcw = np.concatenate([Rc.reshape(-1), tc]).astype(np.float32)
    blob = _search_frame_blob(2, frame, n, m, th, 0.9, scale, angle_cur, state, 645.1, 483.9)
    blob += Tcw.tobytes() + np.array([fx, fy, cx, cy], np.float32).tobytes() + struct.pack("<i", orbdist)
    blob += world.tobytes() + kang.tobytes() + maxd.tobytes() + mind.tobytes() + has.tobytes() + bad.tobytes() + found.tobytes() + level.tobytes()
    blob += qdesc.tobytes()
    pin, pout = tmp_path / "s.bin", tmp_path / "o.bin"
    open(pin, "wb").write(blob)
    subprocess.check_call([EXE, "search", str(pin), str(pout)])
    r = np.fromfile(pout, np.int32)
    assert r[0] == 0
    xc = _gemm_row(Rc[0], tc[0], X, Y, Z)
    yc = _gemm_row(Rc[1], tc[1], X, Y, Z)
    zc = _gemm_row(Rc[2], tc[2], X, Y, Z)
    invz = (1.0 / zc.astype(np.float64)).astype(np.float32)
    u, v = (fx * xc) * invz + cx, (fy * yc) * invz + cy
    ok = (has != 0) & (bad == 0) & (found == 0) & ~(u < f32(frame["min_x"])) & ~(u > f32(645.1)) & ~(v < f32(frame["min_y"])) & ~(v > f32(483.9))
    ok &= ~((d3 < mind) | (d3 > maxd))
    keep = np.nonzero(ok)[0]
    assert 300 < len(keep) < m
    q = dict(uvr=np.stack([u, v, f32(th) * scale[level]], 1)[keep], level_min=level[keep] - 1, level_max=level[keep] + 1, desc=qdesc[keep])
    match, nm, _ = ob.guided_search(dict(frame, kp_taken=(state > 0).astype(np.uint8)), q, orbdist, False, 0.9)
    full = np.full(m, -1, np.int32)
    full[keep] = match
    owner = _apply_in_order(n, state, full)
    hist = [[] for _ in range(30)]
    factor = f32(1.0) / f32(30)
    for qi in keep:
        k = full[qi]
        if k < 0:
            continue
        rot = f32(kang[qi] - angle_cur[k])
        if rot < 0:
            rot = f32(rot + f32(360))
        b = int(np.floor(f32(rot * factor) + f32(0.5)))
        hist[0 if b == 30 else b].append(k)
    sizes = [len(h) for h in hist]
    order = sorted(range(30), key=lambda i: -sizes[i])
    assert sizes[order[0]] > 100 and sizes[order[1]] < 0.1 * sizes[order[0]], "test premise: one dominant rotation bin"
    for i in range(30):
        if i != order[0]:
            for k in hist[i]:
                owner[k] = -1
                nm -= 1
    assert r[1] == nm and np.array_equal(r[2:], owner)


def _ba_blob(prob):
    K, P, E = len(prob["kf_fixed"]), len(prob["pt_xyz"]), len(prob["edge_kf"])
    blob = struct.pack("<iii", K, P, E)
    blob += prob["kf_pose"].astype(np.float32).tobytes()
    blob += prob["kf_fixed"].tobytes() + b"\0" * ((4 - K % 4) % 4)
    blob += prob["kf_intr"][0].astype(np.float32).tobytes()
    blob += prob["pt_xyz"].astype(np.float32).tobytes()
    blob += prob["edge_kf"].astype(np.int32).tobytes() + prob["edge_pt"].astype(np.int32).tobytes()
    blob += prob["edge_uv"].astype(np.float32).tobytes() + prob["edge_inv_sigma2"].astype(np.float32).tobytes()
    return blob, K, P


@pytest.mark.gpu
@pytest.mark.parametrize("name,its,loopkf", [("global_init", 20, 0), ("global_map", 10, 0), ("global_map", 10, 7)])
def test_shim_global_bundle_adjustment(tmp_path, name, its, loopkf):
    """Optimizer::GlobalBundleAdjustemnt(pMap, nIterations, pbStopFlag, nLoopKF, bRobust) through the template: results equal
    the reference g2o's golden run of the same single-stage schedule; nLoopKF != 0 writes mTcwGBA / mPosGBA instead."""
    _build()
    prob, ref = load_ba_golden(os.path.join(ROOT, "tests", "golden", "ba_%s.npz" % name))
    assert ref["schedule"][0] == its and ref["schedule"][1] == 0
    blob, K, P = _ba_blob(prob)
    pin, pout = tmp_path / "p.bin", tmp_path / "o.bin"
    open(pin, "wb").write(blob)
    subprocess.check_call([EXE, "ba", str(pin), str(pout), "global", str(its), str(loopkf), "1"])
    raw = open(pout, "rb").read()
    f = np.frombuffer(raw, np.float32, 12 * K + 3 * P)
    R, t, pts = f[:9 * K].reshape(K, 9), f[9 * K:12 * K].reshape(K, 3), f[12 * K:].reshape(P, 3)
    erased, updates = struct.unpack_from("<ii", raw, 4 * (12 * K + 3 * P))
    tol = 2e-5 if name == "global_init" else 1e-5   # float32 write-back; the 2-keyframe map is scale-weak
    assert np.abs(R - ref["kf_pose"][:, :9]).max() < 2e-6
    assert np.abs(t - ref["kf_pose"][:, 9:]).max() < tol * max(np.abs(ref["kf_pose"][:, 9:]).max(), 1)
    assert np.abs(pts - ref["pt_xyz"]).max() < tol * np.abs(ref["pt_xyz"]).max()
    assert erased == 0                                  # a global BA never erases observations
    assert updates == (P if loopkf == 0 else -P)        # UpdateNormalAndDepth only on the direct write-back path


def _three_maxima(sizes):
    """ORBmatcher::ComputeThreeMaxima (ORBmatcher.cc:1605-1646): indices of the kept bins."""
    max1 = max2 = max3 = 0
    i1 = i2 = i3 = -1
    for i, s in enumerate(sizes):
        if s > max1:
            max3, max2, max1 = max2, max1, s
            i3, i2, i1 = i2, i1, i
        elif s > max2:
            max3, max2 = max2, s
            i3, i2 = i2, i
        elif s > max3:
            max3, i3 = s, i
    if max2 < 0.1 * float(max1):
        i2 = i3 = -1
    elif max3 < 0.1 * float(max1):
        i3 = -1
    return {i1, i2, i3}


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1, 2])
def test_shim_bow_drivers(tmp_path, variant):
    """SearchByBoW(KeyFrame*, Frame&), SearchByBoW(KeyFrame*, KeyFrame*) and SearchForTriangulation through the templates:
    feature-vector walk and rotation histogram on the host, the node loops in one slamit_bow_search call.  Expected =
    the oracle's sequential loops + the same histogram in numpy."""
    from oracle import bindings as ob
    from weiner_slamit_v2_amd import synth

    _build()
    f32 = np.float32
    n1, n2, nodes, nnratio = 1100, 1000, 90, 0.75
    s1, s2, g, epi = synth.synth_bow(n1, n2, nodes, 21 + variant, mode=1)
    rs = np.random.RandomState(17 + variant)
    # map-point state per feature: 0 none, 1 good, 2 bad
    pr = [0.8, 0.15, 0.05] if variant == 2 else [0.3, 0.6, 0.1]   # triangulation matches the features WITHOUT a map point
    mp1 = rs.choice([0, 1, 2], n1, p=pr).astype(np.int32)
    mp2 = rs.choice([0, 1, 2], n2, p=pr).astype(np.int32)
    # the node of every feature, recovered from the groups (features outside common nodes get private node ids)
    node1, node2 = np.full(n1, -1, np.int32), np.full(n2, -1, np.int32)
    for k in range(len(g["q_ptr"]) - 1):
        node1[g["q_idx"][g["q_ptr"][k]:g["q_ptr"][k + 1]]] = 2 * k
        node2[g["c_idx"][g["c_ptr"][k]:g["c_ptr"][k + 1]]] = 2 * k
    node1[node1 < 0] = 2 * np.arange((node1 < 0).sum()) + 1 + 2 * len(g["q_ptr"])
    node2[node2 < 0] = 2 * np.arange((node2 < 0).sum()) + 100001
    # the reference's validity rules per driver
    if variant == 0:
        v1, v2, kw = (mp1 == 1), None, dict(mode=0, th=50, th_inclusive=True, nnratio=nnratio)
    elif variant == 1:
        v1, v2, kw = (mp1 == 1), (mp2 == 1), dict(mode=0, th=50, th_inclusive=False, nnratio=nnratio)
    else:
        v1, v2, kw = (mp1 == 0), (mp2 == 0), dict(mode=1, th=50, epi=epi)
    o1 = dict(s1, valid=v1.astype(np.uint8))
    o2 = dict(s2, valid=None if v2 is None else v2.astype(np.uint8))
    m12, d12, nm = ob.bow_search(o1, o2, g, **kw)
    hit = np.nonzero(m12 >= 0)[0]
    assert len(hit) > 100, "test premise: enough matches for a histogram"
    # angles: one dominant rotation for ~80 % of the matches
    a2 = rs.uniform(0, 360, n2).astype(f32)
    a1 = rs.uniform(0, 360, n1).astype(f32)
    a1[hit] = ((a2[m12[hit]] + np.where(rs.rand(len(hit)) < 0.8, 25.0, rs.uniform(0, 360, len(hit)))) % 360.0).astype(f32)
    # camera geometry that reproduces epi["ex"], epi["ey"]: K2 = (fx, fy, cx, cy), R2w = I, t2w = 0, Cw chosen accordingly
    fx, fy, cx, cy = f32(517.3), f32(516.5), f32(318.6), f32(255.3)
    Cz = f32(2.0)
    Cw = np.array([(f32(epi["ex"]) - cx) / fx * Cz, (f32(epi["ey"]) - cy) / fy * Cz, Cz], f32)
    ex = f32(f32(fx * Cw[0]) * f32(f32(1.0) / Cw[2])) + cx
    ey = f32(f32(fy * Cw[1]) * f32(f32(1.0) / Cw[2])) + cy
    if variant == 2:   # the epipole the shim will compute may differ from the requested one in the last bit: use ITS value
        m12, d12, nm = ob.bow_search(o1, o2, g, **dict(kw, epi=dict(epi, ex=float(ex), ey=float(ey))))
        hit = np.nonzero(m12 >= 0)[0]
        a1[hit] = ((a2[m12[hit]] + np.where(rs.rand(len(hit)) < 0.8, 25.0, rs.uniform(0, 360, len(hit)))) % 360.0).astype(f32)

    def side(s, ang, node, mp, n):
        xy = np.asarray(s["kp_xy"], f32).reshape(n, 2)
        octv = np.asarray(s.get("kp_octave", np.zeros(n)), np.int32)
        return np.asarray(s["desc"], np.uint8).tobytes() + ang.tobytes() + node.astype(np.int32).tobytes() + mp.tobytes() + xy.tobytes() + octv.tobytes()

    blob = struct.pack("<iiif", variant, n1, n2, nnratio) + side(s1, a1, node1, mp1, n1) + side(s2, a2, node2, mp2, n2)
    blob += np.asarray(epi["F12"], f32).tobytes() + Cw.tobytes() + np.eye(3, dtype=f32).tobytes() + np.zeros(3, f32).tobytes()
    blob += np.array([fx, fy, cx, cy], f32).tobytes() + np.asarray(epi["scale_factor"][:8], f32).tobytes() + np.asarray(epi["level_sigma2"][:8], f32).tobytes()
    pin, pout = tmp_path / "b.bin", tmp_path / "o.bin"
    open(pin, "wb").write(blob)
    subprocess.check_call([EXE, "bow", str(pin), str(pout)])
    raw = open(pout, "rb").read()
    status, got_nm = struct.unpack_from("<ii", raw, 0)
    assert status == 0
    # expected: rotation histogram over the matches, keep the three maxima
    hist = [[] for _ in range(30)]
    factor = f32(1.0) / f32(30)
    for i1 in hit:
        rot = f32(a1[i1] - a2[m12[i1]])
        if rot < 0:
            rot = f32(rot + f32(360))
        b = int(np.floor(f32(rot * factor) + f32(0.5)))
        hist[0 if b == 30 else b].append(i1)
    keep = _three_maxima([len(h) for h in hist])
    e12 = m12.copy()
    for i in range(30):
        if i not in keep:
            e12[hist[i]] = -1
    enm = int((e12 >= 0).sum())
    assert enm < len(hit), "test premise: the histogram removes something"
    assert got_nm == enm
    if variant == 0:
        got = np.frombuffer(raw, np.int32, n2, 8)
        exp = np.full(n2, -1, np.int32)
        ok = np.nonzero(e12 >= 0)[0]
        exp[e12[ok]] = ok
        assert np.array_equal(got, exp)
    elif variant == 1:
        assert np.array_equal(np.frombuffer(raw, np.int32, n1, 8), e12)
    else:
        npairs = struct.unpack_from("<i", raw, 8)[0]
        pairs = np.frombuffer(raw, np.int32, 2 * npairs, 12).reshape(npairs, 2)
        ok = np.nonzero(e12 >= 0)[0]
        assert npairs == enm and np.array_equal(pairs[:, 0], ok) and np.array_equal(pairs[:, 1], e12[ok])


# ---- the Sim3 drivers (loop closing) through the templates -----------------------------------------------------------
def _f32_apply(R, t, P):
    """R * p + t the way the shim writes it (float, left to right)."""
    return np.stack([((R[r, 0] * P[:, 0] + R[r, 1] * P[:, 1]) + R[r, 2] * P[:, 2]) + t[r] for r in range(3)], 1).astype(np.float32)


def _norm3(V):
    return np.sqrt((V.astype(np.float64) ** 2).sum(1)).astype(np.float32)   # cv::norm: double accumulation


def _decompose_scw(S):
    f32 = np.float32
    scw = f32(np.sqrt((S[0, :3].astype(np.float64) ** 2).sum()))
    R = (S[:3, :3] / scw).astype(f32)
    t = (S[:3, 3] / scw).astype(f32)
    O = np.array([-(R[:, r].astype(np.float64) * t.astype(np.float64)).sum() for r in range(3)]).astype(f32)
    return R, t, O


def _sim3_scene(seed, n=1300, m=800):
    """A keyframe (synth_search keypoints + grid) and m map points that project near some of its keypoints under (Rc, tc)."""
    from weiner_slamit_v2_amd import synth
    f32 = np.float32
    rs = np.random.RandomState(seed)
    frame, _ = synth.synth_search(n, 4, seed)
    fx, fy, cx, cy = f32(526.69), f32(540.36), f32(313.07), f32(238.39)
    Rc, tc = synth.se3_exp(np.array([0.02, -0.015, 0.03, 0.1, -0.05, 0.02]) * (1 + seed % 3))
    src = rs.randint(0, n, m)
    depth = rs.uniform(2, 8, m)
    px = frame["kp_xy"][src].astype(np.float64) + rs.normal(0, 1.0, (m, 2))
    pc = np.stack([(px[:, 0] - cx) / fx * depth, (px[:, 1] - cy) / fy * depth, depth], 1)
    pc[::41] *= -1                                   # behind the camera
    pos = ((pc - tc) @ Rc).astype(f32)               # world position: Rc^T (pc - tc)
    Ow = -(Rc.T @ tc)
    PO = pos.astype(np.float64) - Ow
    d3 = np.sqrt((PO ** 2).sum(1))
    normal = (PO / np.maximum(d3[:, None], 1e-6)).astype(f32)
    normal[::17] *= -1                               # viewing angle check fails
    maxd, mind = (d3 * 1.5).astype(f32), (d3 * 0.6).astype(f32)
    mind[::23] = (d3[::23] * 1.2).astype(f32)        # outside the scale-invariance range
    level = np.clip(frame["kp_octave"][src] + rs.randint(0, 2, m), 0, 7).astype(np.int32)
    bad = (rs.rand(m) < 0.05).astype(np.int32)
    qdesc = frame["desc"][src].copy()
    for j in range(m):
        for b in rs.randint(0, 256, rs.randint(0, 30)):
            qdesc[j, b >> 3] ^= np.uint8(1 << (b & 7))
    pts = dict(pos=pos, normal=normal, maxd=maxd, mind=mind, level=level, bad=bad, desc=qdesc, src=src)
    cam = dict(R=Rc.astype(f32), t=tc.astype(f32), intr=np.array([fx, fy, cx, cy], f32),
               bounds=np.array([frame["min_x"], 645.1, frame["min_y"], 483.9, frame["inv_w"], frame["inv_h"]], f32),
               scale=(f32(1.2) ** np.arange(8, dtype=f32)).astype(f32))
    return frame, cam, pts


def _points_blob(pts, idx2=None):
    m = len(pts["level"])
    idx2 = np.full(m, -1, np.int32) if idx2 is None else idx2.astype(np.int32)
    return (struct.pack("<i", m) + pts["pos"].tobytes() + pts["normal"].tobytes() + pts["maxd"].tobytes() + pts["mind"].tobytes() +
            pts["level"].tobytes() + pts["bad"].astype(np.int32).tobytes() + idx2.tobytes() + pts["desc"].tobytes())


def _kf_blob(frame, cam, mp):
    n = len(frame["kp_octave"])
    return (cam["R"].tobytes() + cam["t"].tobytes() + cam["intr"].tobytes() + cam["bounds"].tobytes() + cam["scale"].tobytes() + struct.pack("<i", n) +
            frame["kp_xy"].astype(np.float32).tobytes() + frame["kp_octave"].astype(np.int32).tobytes() + mp.astype(np.int32).tobytes() + frame["desc"].tobytes())


def _project_checks(cam, pc, P, O, pts, th, viewing=True, pc_for_dist=None, inv_double=False):
    """u, v, radius and the keep-mask of the reference's per-point tests (depth, image, distance range, viewing angle)."""
    f32 = np.float32
    fx, fy, cx, cy = cam["intr"]
    with np.errstate(divide="ignore", invalid="ignore"):
        invz = (1.0 / pc[:, 2].astype(np.float64)).astype(f32) if inv_double else (f32(1) / pc[:, 2]).astype(f32)
    u, v = fx * (pc[:, 0] * invz) + cx, fy * (pc[:, 1] * invz) + cy
    b = cam["bounds"]
    ok = ~(pc[:, 2] < 0) & (u >= b[0]) & (u < b[1]) & (v >= b[2]) & (v < b[3])
    if pc_for_dist is None:
        PO = np.stack([P[:, 0] - O[0], P[:, 1] - O[1], P[:, 2] - O[2]], 1).astype(f32)
        dist = _norm3(PO)
    else:
        dist = _norm3(pc_for_dist)
    ok &= ~((dist < pts["mind"]) | (dist > pts["maxd"]))
    if viewing:
        dot = (PO.astype(np.float64) * pts["normal"].astype(np.float64)).sum(1)
        ok &= ~(dot < 0.5 * dist.astype(np.float64))
    radius = (f32(th) * cam["scale"][pts["level"]]).astype(f32)
    return u.astype(f32), v.astype(f32), radius, ok


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1])
def test_shim_sim3_projection_and_fuse(tmp_path, variant):
    """ORBmatcher::SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) and ORBmatcher::Fuse(pKF, Scw, vpPoints, th,
    vpReplacePoint) through the templates: Scw decomposition, projection and tests on the host, one guided search on the
    device, the reference's bookkeeping in order."""
    from oracle import bindings as ob

    _build()
    f32 = np.float32
    rs = np.random.RandomState(40 + variant)
    frame, cam, pts = _sim3_scene(60 + variant)
    n, m = len(frame["kp_octave"]), len(pts["level"])
    th = 10 if variant == 0 else 4.0
    s = f32(1.7)
    S = np.zeros((3, 4), f32)
    S[:, :3] = s * cam["R"]
    S[:, 3] = s * cam["t"]
    # which map point already sits at a keypoint
    mp = np.where(rs.rand(n) < 0.3, rs.randint(0, m, n), -1).astype(np.int32)
    init = np.where(rs.rand(n) < 0.25, rs.randint(0, m, n), -1).astype(np.int32)       # variant 0: vpMatched on entry
    blob = struct.pack("<iif", variant, 1, float(th)) + _points_blob(pts) + _kf_blob(frame, cam, mp) + S.tobytes()
    if variant == 0:
        blob += init.tobytes()
    pin, pout = tmp_path / "s.bin", tmp_path / "o.bin"
    open(pin, "wb").write(blob)
    subprocess.check_call([EXE, "sim3", str(pin), str(pout)])
    r = np.fromfile(pout, np.int32)
    assert r[0] == 0
    # expected
    R, t, O = _decompose_scw(np.vstack([S, [0, 0, 0, 1]]).astype(f32))
    pc = _f32_apply(R, t, pts["pos"])
    u, v, radius, ok = _project_checks(cam, pc, pts["pos"], O, pts, th, inv_double=(variant == 1))
    ok &= pts["bad"] == 0
    already = set(init[init >= 0].tolist()) if variant == 0 else set(mp[mp >= 0].tolist())
    ok &= ~np.isin(np.arange(m), list(already))
    keep = np.nonzero(ok)[0]
    assert 150 < len(keep) < m
    q = dict(uvr=np.stack([u, v, radius], 1)[keep], level_min=pts["level"][keep] - 1, level_max=pts["level"][keep], desc=pts["desc"][keep],
             takes=np.full(len(keep), 1 if variant == 0 else 0, np.uint8))
    taken = (init >= 0).astype(np.uint8) if variant == 0 else np.zeros(n, np.uint8)
    match, nm, _ = ob.guided_search(dict(frame, kp_taken=taken), q, 50, False, 0.75)
    if variant == 0:
        exp = init.copy()
        hit = match >= 0
        exp[match[hit]] = keep[hit]
        assert r[1] == int(hit.sum()) and r[1] > 50
        assert np.array_equal(r[2:2 + n], exp)
    else:
        owner = mp.copy()
        replace, added = np.full(m, -1, np.int32), np.full(m, -1, np.int32)
        nf = 0
        for k, j in enumerate(keep):
            bi = match[k]
            if bi < 0:
                continue
            if owner[bi] >= 0:
                if not pts["bad"][owner[bi]]:
                    replace[j] = owner[bi]
            else:
                added[j], owner[bi] = bi, j
            nf += 1
        assert r[1] == nf and nf > 50 and (replace >= 0).sum() > 5 and (added >= 0).sum() > 5
        assert np.array_equal(r[2:2 + m], replace) and np.array_equal(r[2 + m:2 + 2 * m], added) and np.array_equal(r[2 + 2 * m:2 + 2 * m + n], owner)


@pytest.mark.gpu
def test_shim_search_by_sim3(tmp_path):
    """ORBmatcher::SearchBySim3 through the template: both directions are one guided search each, then the agreement test."""
    from oracle import bindings as ob
    from weiner_slamit_v2_amd import synth

    _build()
    f32 = np.float32
    rs = np.random.RandomState(77)
    frame1, cam1, pts = _sim3_scene(71)
    n1, m = len(frame1["kp_octave"]), len(pts["level"])
    th = 7.5
    # camera 2: a second pose; its keypoints are the projections of the map points (+ noise) and some random ones
    R2, t2 = synth.se3_exp(np.array([-0.03, 0.02, 0.01, -0.2, 0.05, 0.1]))
    cam2 = dict(cam1, R=R2.astype(f32), t=t2.astype(f32))
    pc2 = (pts["pos"].astype(np.float64) @ R2.T) + t2
    fx, fy, cx, cy = [float(x) for x in cam1["intr"]]
    with np.errstate(divide="ignore", invalid="ignore"):
        uv2 = np.stack([fx * pc2[:, 0] / pc2[:, 2] + cx, fy * pc2[:, 1] / pc2[:, 2] + cy], 1)
    vis = (pc2[:, 2] > 0.5) & (uv2[:, 0] > 5) & (uv2[:, 0] < 635) & (uv2[:, 1] > 5) & (uv2[:, 1] < 475)
    vidx = np.nonzero(vis)[0]
    extra = 400
    xy2 = np.concatenate([uv2[vidx] + rs.normal(0, 0.8, (len(vidx), 2)), np.stack([rs.uniform(0, 640, extra), rs.uniform(0, 480, extra)], 1)]).astype(f32)
    oct2 = np.concatenate([np.clip(pts["level"][vidx] - rs.randint(0, 2, len(vidx)), 0, 7), rs.randint(0, 8, extra)]).astype(np.int32)
    d2 = np.concatenate([pts["desc"][vidx], rs.randint(0, 256, (extra, 32)).astype(np.uint8)])
    for i in range(len(vidx)):
        for b in rs.randint(0, 256, rs.randint(0, 40)):
            d2[i, b >> 3] ^= np.uint8(1 << (b & 7))
    n2 = len(xy2)
    frame2 = dict(frame1, kp_xy=xy2, kp_octave=oct2, desc=d2)
    # map points at keypoints: KF1 keypoint src[j] holds point j (last writer wins), KF2 keypoint i holds point vidx[i]
    mp1 = np.full(n1, -1, np.int32)
    mp1[pts["src"]] = np.arange(m)
    mp2 = np.full(n2, -1, np.int32)
    mp2[:len(vidx)] = vidx
    mp2[rs.rand(n2) < 0.1] = -1
    idx_in_kf2 = np.full(m, -1, np.int32)
    have = np.nonzero(mp2 >= 0)[0]
    idx_in_kf2[mp2[have]] = have
    init = np.where((rs.rand(n1) < 0.1) & (mp1 >= 0), mp1, -1).astype(np.int32)      # matches known on entry
    # similarity between the cameras: p_c1 = s12 R12 p_c2 + t12 (s12 slightly off 1: projections move a little)
    s12 = f32(1.01)
    R12 = (cam1["R"].astype(np.float64) @ R2.T).astype(f32)
    t12 = (cam1["t"].astype(np.float64) - R12.astype(np.float64) @ t2).astype(f32)
    blob = struct.pack("<iif", 2, 2, th) + _points_blob(pts, idx_in_kf2) + _kf_blob(frame1, cam1, mp1) + _kf_blob(frame2, cam2, mp2)
    blob += struct.pack("<f", float(s12)) + R12.tobytes() + t12.tobytes() + init.tobytes()
    pin, pout = tmp_path / "s.bin", tmp_path / "o.bin"
    open(pin, "wb").write(blob)
    subprocess.check_call([EXE, "sim3", str(pin), str(pout)])
    r = np.fromfile(pout, np.int32)
    assert r[0] == 0
    # expected (ORBmatcher.cc:1119-1121, 1146-1323)
    sR12 = (s12 * R12).astype(f32)
    sR21 = ((1.0 / float(s12)) * R12.T.astype(np.float64)).astype(f32)
    t21 = np.array([-(sR21[r_].astype(np.float64) * t12.astype(np.float64)).sum() for r_ in range(3)]).astype(f32)
    done1 = init >= 0
    done2 = np.zeros(n2, bool)
    for i in np.nonzero(done1)[0]:
        k = idx_in_kf2[init[i]]
        if 0 <= k < n2:
            done2[k] = True

    def direction(mp_from, done, cam_from, sR, ts, frame_to, cam_to):
        ids = np.nonzero((mp_from >= 0) & ~done)[0]
        pid = mp_from[ids]
        sub = {k: pts[k][pid] for k in ("pos", "normal", "maxd", "mind", "level", "bad", "desc")}
        pa = _f32_apply(cam_from["R"], cam_from["t"], sub["pos"])
        pb = _f32_apply(sR, ts, pa)
        camx = dict(cam_to, intr=cam1["intr"])                      # both directions use camera 1's intrinsics
        u, v, radius, ok = _project_checks(camx, pb, None, None, sub, th, viewing=False, pc_for_dist=pb, inv_double=True)
        ok &= sub["bad"] == 0
        keep = np.nonzero(ok)[0]
        q = dict(uvr=np.stack([u, v, radius], 1)[keep], level_min=sub["level"][keep] - 1, level_max=sub["level"][keep], desc=sub["desc"][keep],
                 takes=np.zeros(len(keep), np.uint8))
        match, _, _ = ob.guided_search(dict(frame_to, kp_taken=np.zeros(len(frame_to["kp_octave"]), np.uint8)), q, 100, False, 0.75)
        out = np.full(len(mp_from), -1, np.int32)
        out[ids[keep]] = match
        return out

    vn1 = direction(mp1, done1, cam1, sR21, t21, frame2, cam2)
    vn2 = direction(mp2, done2, cam2, sR12, t12, frame1, cam1)
    exp = init.copy()
    nfound = 0
    for i1 in range(n1):
        k = vn1[i1]
        if k >= 0 and vn2[k] == i1:
            exp[i1] = mp2[k]
            nfound += 1
    assert nfound > 100 and (vn1 >= 0).sum() > nfound, "test premise: matches, and some that fail the agreement test"
    assert r[1] == nfound
    assert np.array_equal(r[2:2 + n1], exp)


@pytest.mark.gpu
@pytest.mark.parametrize("fix_scale", [False, True])
def test_shim_optimize_sim3(tmp_path, fix_scale):
    """Optimizer::OptimizeSim3 through the template: validity tests and camera-frame points on the host, the two-stage
    optimisation in one device call, vpMatches1 / g2oS12 written back.  Expected = the same problem assembled in numpy and
    solved by the CPU oracle (itself pinned to the reference's g2o)."""
    from oracle import bindings as ob
    from weiner_slamit_v2_amd import synth

    _build()
    f32 = np.float32
    rs = np.random.RandomState(91 + fix_scale)
    npairs, extra = 260, 140
    # two keyframes looking at the same points; KF2's map points are a scaled / shifted copy (a drifted loop)
    R1, t1 = synth.se3_exp(np.array([0.02, -0.03, 0.01, 0.1, 0.05, -0.02]))
    R2, t2 = synth.se3_exp(np.array([-0.04, 0.05, 0.02, -0.3, 0.1, 0.15]))
    s_true = 1.0 if fix_scale else 1.07
    Pw1 = np.stack([rs.uniform(-2.5, 2.5, npairs), rs.uniform(-1.8, 1.8, npairs), rs.uniform(3.0, 9.0, npairs)], 1)
    Pw1 = (Pw1 - t1) @ R1                                   # world positions of KF1's points (seen in front of camera 1)
    pc1 = Pw1 @ R1.T + t1
    Rd_, td_ = synth.se3_exp(np.array([0.03, 0.02, -0.04, 0.2, -0.1, 0.05]))
    pc2 = ((pc1 - td_) @ Rd_) / s_true                      # p_c1 = s R p_c2 + t  with (R, t) = (Rd_, td_)
    Pw2 = (pc2 - t2) @ R2 + rs.normal(0, 0.01, (npairs, 3))  # KF2's own estimate of the same points
    K1 = np.array([517.3, 516.5, 318.6, 255.3], f32)
    K2 = np.array([520.9, 521.0, 325.1, 249.7], f32)
    pos = np.concatenate([Pw1, Pw2]).astype(f32)            # map points 0..npairs-1 belong to KF1, npairs.. to KF2
    m = len(pos)
    cams = [(R1.astype(f32), t1.astype(f32), K1), (R2.astype(f32), t2.astype(f32), K2)]

    def cam_points(k, P):                                    # R * P + t as cv::gemm computes it (double sums, float result)
        Rk, tk, _ = cams[k]
        return (P.astype(np.float64) @ Rk.astype(np.float64).T + tk.astype(np.float64)).astype(f32)

    def project(k, pc):
        Kk = cams[k][2].astype(np.float64)
        return np.stack([Kk[0] * pc[:, 0] / pc[:, 2] + Kk[2], Kk[1] * pc[:, 1] / pc[:, 2] + Kk[3]], 1)

    n1, n2 = npairs + extra, npairs + extra
    perm1, perm2 = rs.permutation(n1)[:npairs], rs.permutation(n2)[:npairs]          # keypoint index of pair j in each keyframe
    xy1 = np.stack([rs.uniform(0, 640, n1), rs.uniform(0, 480, n1)], 1)
    xy2 = np.stack([rs.uniform(0, 640, n2), rs.uniform(0, 480, n2)], 1)
    xy1[perm1] = project(0, cam_points(0, pos[:npairs]).astype(np.float64)) + rs.normal(0, 0.7, (npairs, 2))
    xy2[perm2] = project(1, cam_points(1, pos[npairs:]).astype(np.float64)) + rs.normal(0, 0.7, (npairs, 2))
    wrong = rs.rand(npairs) < 0.2
    xy1[perm1[wrong]] += rs.uniform(-50, 50, (int(wrong.sum()), 2))
    xy1, xy2 = xy1.astype(f32), xy2.astype(f32)
    oct1, oct2 = rs.randint(0, 8, n1).astype(np.int32), rs.randint(0, 8, n2).astype(np.int32)
    scale = f32(1.2) ** np.arange(8, dtype=f32)
    invsig = (f32(1) / (scale * scale)).astype(f32)
    mp1, mp2 = np.full(n1, -1, np.int32), np.full(n2, -1, np.int32)
    mp1[perm1] = np.arange(npairs)
    mp2[perm2] = npairs + np.arange(npairs)
    bad = (rs.rand(m) < 0.04).astype(np.int32)
    idx_in_kf2 = np.full(m, -1, np.int32)
    idx_in_kf2[npairs:] = perm2
    idx_in_kf2[npairs:][rs.rand(npairs) < 0.03] = -1                                   # the matched point is not (any more) in KF2
    matches1 = np.full(n1, -1, np.int32)
    matches1[perm1] = npairs + np.arange(npairs)
    matches1[perm1[rs.rand(npairs) < 0.1]] = -1                                        # no match for this keypoint
    mp1[perm1[rs.rand(npairs) < 0.03]] = -1                                            # KF1 has no point there
    # initial S12: perturbed truth
    dR, dt = synth.se3_exp(rs.normal(0, 0.02, 6))
    R0, t0, s0 = dR @ Rd_, dR @ td_ + dt, s_true * (1.0 if fix_scale else 1.03)
    S12 = np.concatenate([R0.reshape(9), t0, [s0]])
    th2 = 10.0
    blob = struct.pack("<iiiif", n1, n2, m, int(fix_scale), th2) + S12.astype(np.float64).tobytes()
    for (Rk, tk, Kk), xy, octv, mp in ((cams[0], xy1, oct1, mp1), (cams[1], xy2, oct2, mp2)):
        blob += Kk.tobytes() + Rk.tobytes() + tk.tobytes() + invsig.tobytes() + xy.tobytes() + octv.tobytes() + mp.tobytes()
    blob += pos.tobytes() + bad.tobytes() + idx_in_kf2.tobytes() + matches1.tobytes()
    pin, pout = tmp_path / "s.bin", tmp_path / "o.bin"
    open(pin, "wb").write(blob)
    subprocess.check_call([EXE, "osim3", str(pin), str(pout)])
    raw = open(pout, "rb").read()
    status, nin = struct.unpack_from("<ii", raw, 0)
    got_m = np.frombuffer(raw, np.int32, n1, 8)
    got_S = np.frombuffer(raw, np.float64, 13, 8 + 4 * n1)
    assert status == 0
    # expected: the reference's validity tests (Optimizer.cc:1099-1136), then the oracle
    idx = [i for i in range(n1) if matches1[i] >= 0 and mp1[i] >= 0 and not bad[mp1[i]] and not bad[matches1[i]] and idx_in_kf2[matches1[i]] >= 0]
    idx = np.array(idx)
    assert 150 < len(idx) < npairs
    j1, j2 = mp1[idx], matches1[idx]
    i2 = idx_in_kf2[j2]
    prob = dict(p1=cam_points(0, pos[j1]).astype(np.float64), p2=cam_points(1, pos[j2]).astype(np.float64), obs1=xy1[idx].astype(np.float64),
                obs2=xy2[i2].astype(np.float64), inv_sigma2_1=invsig[oct1[idx]].astype(np.float64), inv_sigma2_2=invsig[oct2[i2]].astype(np.float64),
                intr1=K1.astype(np.float64), intr2=K2.astype(np.float64), r12=R0.reshape(9), t12=t0, s12=s0, th2=th2, fix_scale=int(fix_scale))
    o = ob.sim3_solve(prob)
    assert o["n_inliers"] > 100 and (o["inlier"] == 0).sum() > 10
    exp_m = matches1.copy()
    exp_m[idx[o["inlier"] == 0]] = -1
    assert nin == o["n_inliers"] and np.array_equal(got_m, exp_m)
    assert np.abs(got_S[:9] - o["r12"].reshape(9)).max() < 1e-6 and np.abs(got_S[9:12] - o["t12"]).max() < 1e-6 and abs(got_S[12] - o["s12"]) < 1e-6
    assert abs(o["s12"] - s_true) < 0.01
