"""C-ABI behaviour on the GPU: error codes instead of crashes, independent handles on two threads
(the reference runs two extractors on two threads for stereo, Frame.cc:93-94), re-entrant matcher."""
import ctypes as C
import threading

import numpy as np
import pytest

from oracle import bindings as ob
from weiner_slamit_v2_amd import api, synth

pytestmark = pytest.mark.gpu


def test_error_codes_not_crashes():
    L = api.lib()
    h = C.c_void_p()
    assert L.slamit_orb_create(None, 0, C.byref(h)) == -1
    bad = api.OrbParams(1000, 1.2, 99, 20, 7, 640, 480, 1)        # too many levels
    assert L.slamit_orb_create(C.byref(bad), 0, C.byref(h)) == -1 and b"range" in L.slamit_last_error()
    tiny = api.OrbParams(1000, 1.2, 8, 20, 7, 100, 80, 1)          # top levels smaller than one FAST cell
    assert L.slamit_orb_create(C.byref(tiny), 0, C.byref(h)) == -1
    tall = api.OrbParams(1000, 1.2, 2, 20, 7, 100, 400, 1)         # octree needs width >= height/2
    assert L.slamit_orb_create(C.byref(tall), 0, C.byref(h)) == -1
    ok = api.OrbParams(1000, 1.2, 8, 20, 7, 640, 480, 2)
    assert L.slamit_orb_create(C.byref(ok), 0, C.byref(h)) == 0
    img = synth.synth_frame(640, 480, 3)
    cap = L.slamit_orb_max_keypoints(h)
    kps = np.zeros(cap, api.KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = C.c_int()
    assert L.slamit_orb_extract(h, img.ctypes.data, 640, kps.ctypes.data, desc.ctypes.data, 10, C.byref(n)) == -3  # cap too small
    assert L.slamit_orb_extract_batch(h, img.ctypes.data, 640, 640 * 480, 3, kps.ctypes.data, desc.ctypes.data, cap, C.byref(n)) == -3
    assert L.slamit_orb_level(h, 0, 0, None, 0, None, None) == 0
    buf = np.zeros(16, np.uint8)
    assert L.slamit_orb_level(h, 0, 0, buf.ctypes.data, 16, None, None) == -4                # nothing extracted yet
    assert L.slamit_orb_extract(h, img.ctypes.data, 640, kps.ctypes.data, desc.ctypes.data, cap, C.byref(n)) == 0 and n.value >= 1000
    assert L.slamit_orb_level(h, 0, 0, buf.ctypes.data, 16, None, None) == -3                # dst too small
    assert L.slamit_orb_level(h, 0, 9, None, 0, None, None) == -1
    L.slamit_orb_destroy(h)
    L.slamit_orb_destroy(None)                                                                   # harmless
    hb = C.c_void_p()
    assert L.slamit_ba_create(0, 10, 10, 1, 0, C.byref(hb)) == -1
    assert L.slamit_ba_create(4, 10, 10, 1, 0, C.byref(hb)) == 0
    prob = synth.synth_ba(6, 20, 2, seed=1)                                                      # 6 KFs > capacity 4
    p, keep = api._ba_problem(prob)
    o = api.BaOpts(5, 10, api.HUBER_MONO, 5.991, None)
    r, out, st = api.Optimizer._result(p.n_kf, p.n_pt, p.n_edge)
    assert L.slamit_ba_solve(hb, C.byref(p), C.byref(o), C.byref(r)) == -3
    L.slamit_ba_destroy(hb)
    bad_edges = dict(synth.synth_ba(3, 10, 2, seed=2))
    bad_edges["edge_kf"] = bad_edges["edge_kf"].copy()
    bad_edges["edge_kf"][0] = 7
    with pytest.raises(api.SlamitError):
        api.Optimizer(8, 16, 64).LocalBundleAdjustment(bad_edges)


def test_two_extractors_on_two_threads():
    imgs = [synth.synth_frame(640, 480, 50), synth.synth_frame(640, 480, 51)]
    exts = [api.ORBextractor(1000), api.ORBextractor(1000)]
    res = [None, None]

    def work(i):
        out = None
        for _ in range(5):
            out = exts[i](imgs[i])
        res[i] = out

    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    orc = ob.OrbOracle(1000)
    for i in range(2):
        ko, do = orc.extract(imgs[i])
        assert len(res[i][0]) == len(ko) and np.array_equal(res[i][1], do) and np.array_equal(res[i][0]["x"], ko["x"])


def test_matcher_is_reentrant():
    rs = np.random.RandomState(1)
    q = rs.randint(0, 256, (500, 32)).astype(np.uint8)
    t = rs.randint(0, 256, (700, 32)).astype(np.uint8)
    want = ob.best2(q, t)
    got = [None] * 4

    def work(i):
        for _ in range(10):
            got[i] = api.ORBmatcher.best2(q, t)

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for th in ts:
        th.start()
    for th in ts:
        th.join()
    for g in got:
        assert all(np.array_equal(a, b) for a, b in zip(g, want))


def test_full_size_batch_properties():
    """BASELINE-size batch (64 VGA frames): every frame of the batch equals the single-frame result,
    and a second run of the same batch is identical (no state leaks between calls)."""
    frames = np.stack([synth.synth_frame(640, 480, 100 + (i % 8)) for i in range(64)])
    ext = api.ORBextractor(1000, max_batch=64)
    ks, ds = ext.extract_batch(frames)
    single = api.ORBextractor(1000)
    for i in range(8):
        k1, d1 = single(frames[i])
        for j in range(i, 64, 8):
            assert np.array_equal(ks[j], k1) and np.array_equal(ds[j], d1)
    ks2, ds2 = ext.extract_batch(frames)
    assert all(np.array_equal(a, b) for a, b in zip(ds, ds2))


def test_error_codes_of_the_frame_and_search_entry_points():
    L = api.lib()
    cam = api.Camera(500, 500, 320, 240, 0.1, 0, 0, 0, 0)
    xy = np.zeros((4, 2), np.float32)
    assert L.slamit_undistort_points(0, None, xy.ctypes.data, 4, xy.ctypes.data) == -1
    assert L.slamit_undistort_points(0, C.byref(cam), None, 4, xy.ctypes.data) == -1
    assert L.slamit_undistort_points(0, C.byref(cam), None, 0, None) == 0
    start = np.zeros(64 * 48 + 1, np.int32)
    assert L.slamit_frame_finish(0, C.byref(cam), None, 0, 0.0, 0.0, 0.1, 0.1, None, start.ctypes.data, None) == 0 and start[-1] == 0
    assert L.slamit_frame_finish(0, C.byref(cam), None, 5, 0.0, 0.0, 0.1, 0.1, None, start.ctypes.data, None) == -1
    assert L.slamit_frame_finish_batch_dev(0, C.byref(cam), None, None, 10, 2, 0.0, 0.0, 0.1, 0.1, None, None, None, None) == -1
    rule = api._search_rule(100, True, 0.8)
    nm = C.c_int32(7)
    assert L.slamit_guided_search(0, None, None, C.byref(rule), None, C.byref(nm), None, None, None, None) == -1
    fv = api.FrameView(0, None, None, None, None, 0, 0, 0.1, 0.1)
    sq = api.SearchQueries(0, None, None, None, None, None, None)
    assert L.slamit_guided_search(0, C.byref(fv), C.byref(sq), C.byref(rule), None, C.byref(nm), None, None, None, None) == 0 and nm.value == 0
    sq.m = 3   # queries announced but no arrays
    assert L.slamit_guided_search(0, C.byref(fv), C.byref(sq), C.byref(rule), None, C.byref(nm), None, None, None, None) == -1
    assert L.slamit_guided_search_workspace(-1, 10) == 0 and L.slamit_guided_search_workspace(2, 10) > 2 * 10 * 8 * 128
    assert L.slamit_guided_search_batch_dev(0, None, C.byref(rule), None, None, None, None, 0, None) == -1
    # a mismatched array length is caught by the Python binding before it reaches the C-ABI
    f, q = synth.synth_search(50, 20, 1)
    q = dict(q, level_min=q["level_min"][:5])
    with pytest.raises(api.SlamitError):
        api.ORBmatcher.guided_search(f, q)
