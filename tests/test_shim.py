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
def test_shim_optimizer_local_ba(tmp_path):
    _build()
    prob, ref = load_ba_golden(os.path.join(ROOT, "tests", "golden", "ba_fixed3.npz"))
    K, P, E = len(prob["kf_fixed"]), len(prob["pt_xyz"]), len(prob["edge_kf"])
    blob = struct.pack("<iii", K, P, E)
    blob += prob["kf_pose"].astype(np.float32).tobytes()
    blob += prob["kf_fixed"].tobytes() + b"\0" * ((4 - K % 4) % 4)
    blob += prob["kf_intr"][0].astype(np.float32).tobytes()
    blob += prob["pt_xyz"].astype(np.float32).tobytes()
    blob += prob["edge_kf"].astype(np.int32).tobytes() + prob["edge_pt"].astype(np.int32).tobytes()
    blob += prob["edge_uv"].astype(np.float32).tobytes() + prob["edge_inv_sigma2"].astype(np.float32).tobytes()
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
def test_shim_optimizer_pose_optimization(tmp_path):
    _build()
    prob, ref = load_pose_golden(os.path.join(ROOT, "tests", "golden", "pose_typical.npz"))
    n = len(prob["inv_sigma2"])
    blob = struct.pack("<i", n) + prob["pose"].astype(np.float32).tobytes() + prob["intr"].astype(np.float32).tobytes()
    blob += prob["xw"].astype(np.float32).tobytes() + prob["uv"].astype(np.float32).tobytes() + prob["inv_sigma2"].astype(np.float32).tobytes()
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
