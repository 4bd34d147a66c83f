"""Randomised parity stress (not part of the test suite): many seeds of every device path against the oracle."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import bindings as ob  # noqa: E402
from weiner_slamit_v2_amd import api, synth  # noqa: E402

t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 120
bad = []
it = 0
rs = np.random.RandomState(12345)
ext = {}
while time.time() < t_end:
    it += 1
    seed = int(rs.randint(0, 1 << 30))
    # 0. vocabulary-node search (SearchByBoW / SearchForTriangulation loops)
    bmode = int(rs.randint(0, 2))
    b1, b2, bg, bepi = synth.synth_bow(int(rs.randint(0, 2500)), int(rs.randint(0, 2500)), int(rs.randint(1, 300)), seed % 100000, mode=bmode,
                                       big_group=int(rs.choice([0, 0, 0, 300])))
    bkw = dict(mode=bmode, th=int(rs.choice([50, 100])), th_inclusive=bool(rs.rand() < 0.5), nnratio=float(rs.choice([0.6, 0.75, 0.9])), epi=bepi)
    try:
        ga, go = api.ORBmatcher.bow_search(b1, b2, bg, **bkw), ob.bow_search(b1, b2, bg, **bkw)
        if not (np.array_equal(ga[0], go[0]) and np.array_equal(ga[1], go[1]) and ga[2] == go[2]):
            bad.append(("bow", seed, bmode))
    except Exception as e:  # noqa: BLE001
        bad.append(("bow exception", seed, str(e)[:80]))
    # 1. guided search, modes 0 (with / without ratio, gate) and 1
    n, m = int(rs.randint(0, 3000)), int(rs.randint(0, 1500))
    crowd = rs.rand() < 0.2
    f, q = synth.synth_search(n, m, seed % 100000, th=float(rs.choice([1.0, 3.0, 7.0])), crowd=crowd)
    if crowd:
        q["uvr"][:, 2] = np.minimum(q["uvr"][:, 2], 12.0)
    use_ratio, th = bool(rs.rand() < 0.5), int(rs.choice([50, 100]))
    gate = float(rs.choice([0.0, 5.99]))
    sig = (1.0 / (np.float32(1.2) ** np.arange(16, dtype=np.float32)) ** 2).astype(np.float32)
    try:
        g = api.ORBmatcher.guided_search(f, q, th, use_ratio, 0.8, chi2_gate=gate, inv_level_sigma2=sig)
        o = ob.guided_search(f, q, th, use_ratio, 0.8, gate, sig)
        if not (np.array_equal(g[0], o[0]) and g[1] == o[1] and np.array_equal(g[2], o[2])):
            bad.append(("search", seed, n, m))
    except api.SlamitError as e:
        if "MAX_CAND" not in str(e):
            bad.append(("search-err", seed, str(e)))
    if n >= 8:
        f1, prev, f2 = synth.synth_init_pair(n, seed % 1000)
        w = int(rs.choice([10, 40, 100]))
        g = api.ORBmatcher.search_for_initialization(f1, prev, f2, w, 0.9, 50)
        o = ob.search_for_initialization(f1, prev, f2, w, 0.9, 50)
        if not (np.array_equal(g[0], o[0]) and g[1] == o[1] and np.array_equal(g[2], o[2])):
            bad.append(("init", seed, n, w))
    # 2. frame epilogue
    cam = [500 + 60 * rs.rand(), 500 + 60 * rs.rand(), 300 + 40 * rs.rand(), 220 + 40 * rs.rand(),
           float(rs.choice([0.0, 0.1, -0.3, 0.26])), rs.uniform(-1, 1), rs.uniform(-0.01, 0.01), rs.uniform(-0.01, 0.01), float(rs.choice([0.0, 1.1]))]
    kps = np.zeros(int(rs.randint(0, 4000)), api.KP_DTYPE)
    kps["x"], kps["y"] = rs.uniform(0, 640, len(kps)), rs.uniform(0, 480, len(kps))
    b = api.Frame.ComputeImageBounds(cam, 640, 480)
    g = api.Frame.finish(cam, kps, b[0], b[2], b[4], b[5])
    o = ob.frame_finish(cam, kps, b[0], b[2], b[4], b[5])
    if not (np.array_equal(g[0].view(np.uint8), o[0].view(np.uint8)) and np.array_equal(g[1], o[1]) and np.array_equal(g[2], o[2])):
        bad.append(("frame", seed))
    # 3. extractor on a random image kind / geometry
    w_, h_ = int(rs.choice([640, 752, 512, 1024])), int(rs.choice([480, 376, 600]))
    nf = int(rs.choice([500, 1000, 2000]))
    key = (w_, h_, nf)
    if key not in ext:
        ext[key] = (api.ORBextractor(nf, 1.2, 8, 20, 7), ob.OrbOracle(nf))
    kind = rs.randint(0, 3)
    img = synth.synth_frame(w_, h_, seed % 100000) if kind == 0 else synth.noise_frame(w_, h_, seed % 100000) if kind == 1 else \
        ((synth.synth_frame(w_, h_, seed % 100000).astype(np.int32) - 128) // int(rs.choice([2, 5, 9])) + 128).astype(np.uint8)
    kg, dg = ext[key][0](img)
    ko, do = ext[key][1].extract(img)
    if not (len(kg) == len(ko) and np.array_equal(dg, do) and all(np.array_equal(kg[k], ko[k]) for k in ("x", "y", "octave", "response")) and
            np.array_equal(kg["angle"].view(np.uint32), ko["angle"].view(np.uint32))):
        bad.append(("orb", seed, key, kind))
    # 4. matcher
    a_, b_ = rs.randint(0, 256, (int(rs.randint(1, 1500)), 32)).astype(np.uint8), rs.randint(0, 256, (int(rs.randint(0, 1500)), 32)).astype(np.uint8)
    if len(b_) > 10:
        b_[:10] = a_[:1]
    g = api.ORBmatcher.best2(a_, b_)
    o = ob.best2(a_, b_)
    if not all(np.array_equal(x, y) for x, y in zip(g, o)):
        bad.append(("best2", seed))
print("iterations %d, mismatches %d" % (it, len(bad)))
for x in bad[:20]:
    print(x)
sys.exit(1 if bad else 0)
