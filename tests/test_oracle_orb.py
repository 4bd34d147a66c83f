"""Known-answer tests that pin the ORB oracle (oracle/orb_oracle.cc).

The reference holds no test or golden vector for ORBextractor / ORBmatcher and
OpenCV 2.4.9 is not available, so the OpenCV primitives are pinned DEFINITIONALLY: each test
below states the primitive's definition independently (pure Python / numpy, different code
shape from the oracle) and compares.  "parity unpinned" vs a real OpenCV build.
"""
import math

import numpy as np
import pytest

from oracle import bindings as ob
from weiner_slamit_v2_amd import synth

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
        (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


# ---- tables -------------------------------------------------------------------------------

def test_tables_1000():
    t = ob.OrbOracle(1000).tables()
    assert list(t["umax"]) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert list(t["per_level"]) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert t["scale"][0] == 1.0 and abs(t["scale"][7] - 1.2 ** 7) < 1e-5
    assert np.allclose(t["sigma2"], t["scale"] ** 2, rtol=1e-6)
    assert np.allclose(t["inv_sigma2"], 1 / t["sigma2"], rtol=1e-6)


def test_tables_2000():
    assert list(ob.OrbOracle(2000).tables()["per_level"]) == [434, 362, 302, 251, 209, 175, 145, 122]


@pytest.mark.parametrize("wh,expect", [
    ((640, 480), [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]),
    ((1280, 720), [(1280, 720), (1067, 600), (889, 500), (741, 417), (617, 347), (514, 289), (429, 241), (357, 201)]),
])
def test_level_sizes(wh, expect):
    o = ob.OrbOracle(1000)
    o.extract(synth.flat_frame(*wh))
    assert [o.level_size(l) for l in range(8)] == expect


def test_cv_round_half_even():
    assert [ob.cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


# ---- FAST-9/16 -------------------------------------------------------------------------------

def _fast_score_def(img, x, y):
    """max t such that (x,y) is a FAST-9/16 corner at threshold t, or -1."""
    v = int(img[y, x])
    d = [v - int(img[y + dy, x + dx]) for dx, dy in RING]
    best = -1
    for s in range(16):
        arc = [d[(s + k) % 16] for k in range(9)]
        lo = min(arc)   # all ring darker than v by > t  <=> min(v - ring) > t
        hi = -max(arc)  # all ring brighter
        best = max(best, lo - 1, hi - 1)
    return best


def _fast_def(img, t):
    h, w = img.shape
    S = np.zeros((h, w), np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = _fast_score_def(img, x, y)
            if s >= t:
                S[y, x] = s
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = S[y, x]
            if s <= 0 and not (s == 0 and t == 0 and _fast_score_def(img, x, y) >= 0):
                continue
            nb = S[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (s > nb).all():
                out.append((x, y, s))
    return out


@pytest.mark.parametrize("seed", range(6))
def test_fast_vs_definition(seed):
    rs = np.random.RandomState(seed)
    h, w = rs.randint(9, 26), rs.randint(9, 26)
    base = rs.randint(0, 256, size=(h, w))
    if seed % 2:  # blocky image: many real corners and equal-score ties
        base = np.kron(rs.randint(0, 2, size=(h // 4 + 1, w // 4 + 1)) * 120 + 60, np.ones((4, 4)))[:h, :w]
        base = base + rs.randint(-3, 4, size=(h, w))
    img = np.clip(base, 0, 255).astype(np.uint8)
    for t in (7, 20, 40):
        got = [tuple(r) for r in ob.fast(img, t)]
        assert got == _fast_def(img, t)


def test_fast_too_small():
    assert len(ob.fast(np.zeros((6, 30), np.uint8), 20)) == 0
    assert len(ob.fast(np.zeros((30, 6), np.uint8), 20)) == 0


def test_fast_known_corner():
    img = np.full((15, 15), 50, np.uint8)
    img[7:, 7:] = 200  # L-shaped step: the corner pixel sees 11 of 16 ring pixels darker by 150
    assert _fast_score_def(img, 7, 7) == 149
    # its neighbours (8,7), (7,8) score 149 too: strict '>' NMS annihilates equal neighbours
    assert [tuple(r) for r in ob.fast(img, 20)] == _fast_def(img, 20)
    assert (7, 7, 149) not in [tuple(r) for r in ob.fast(img, 20)]
    img[7, 7] = 220  # now the apex is unique
    assert (7, 7, 169) in [tuple(r) for r in ob.fast(img, 20)]


# ---- resize ----------------------------------------------------------------------------------

def _resize_model(src, dw, dh):
    """Independent model of the 8-bit fixed-point bilinear resize using exact rational/IEEE
    steps spelled out with numpy float32/float64 scalars."""
    sh, sw = src.shape

    def coeffs(dn, sn):
        scale = 1.0 / (np.float64(dn) / np.float64(sn))
        ofs, al = [], []
        for d in range(dn):
            f = np.float32((d + 0.5) * scale - 0.5)
            s = int(math.floor(f))
            f = np.float32(f - np.float32(s))
            ofs.append(s)
            al.append(f)
        return ofs, al

    xo, xa = coeffs(dw, sw)
    yo, ya = coeffs(dh, sh)
    out = np.zeros((dh, dw), np.uint8)

    def q(v):  # saturate_cast<short>(float * 2048): round half even
        r = int(np.rint(np.float64(np.float32(v) * np.float32(2048))))
        return max(-32768, min(32767, r))

    for dy in range(dh):
        sy, fy = yo[dy], ya[dy]
        b0, b1 = q(np.float32(1) - fy), q(fy)
        r0 = min(max(sy, 0), sh - 1)
        r1 = min(max(sy + 1, 0), sh - 1)
        for dx in range(dw):
            sx, fx = xo[dx], xa[dx]
            if sx < 0:
                sx, fx = 0, np.float32(0)
            if sx >= sw - 1:
                sx, fx = sw - 1, np.float32(0)
            a0, a1 = q(np.float32(1) - fx), q(fx)
            sx1 = min(sx + 1, sw - 1)
            h0 = int(src[r0, sx]) * a0 + int(src[r0, sx1]) * a1
            h1 = int(src[r1, sx]) * a0 + int(src[r1, sx1]) * a1
            out[dy, dx] = ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2) & 0xFF
    return out


@pytest.mark.parametrize("sw,sh,dw,dh", [(60, 48, 50, 40), (37, 29, 31, 24), (64, 20, 53, 17)])
def test_resize_vs_model(sw, sh, dw, dh):
    src = np.random.RandomState(sw).randint(0, 256, size=(sh, sw)).astype(np.uint8)
    assert np.array_equal(ob.resize(src, dw, dh), _resize_model(src, dw, dh))


def test_resize_constant_and_range():
    src = np.full((48, 64), 255, np.uint8)
    assert (ob.resize(src, 53, 40) == 255).all()
    src[:] = 0
    assert (ob.resize(src, 53, 40) == 0).all()
    g = np.tile(np.arange(64, dtype=np.uint8) * 4, (48, 1))
    r = ob.resize(g, 53, 40)
    # truncating shifts in the vertical pass may lose one count depending on the row weights
    assert (np.diff(r[0].astype(int)) >= 0).all() and (np.abs(r.astype(int) - r[0].astype(int)) <= 1).all()


# ---- Gaussian blur -----------------------------------------------------------------------------

def test_gauss_taps():
    assert list(ob.gauss_taps()) == [18, 34, 49, 55, 49, 34, 18]  # SURVEY Appendix A.4, sum 257


def _blur_model(src):
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    h, w = src.shape

    def refl(p, n):
        while p < 0 or p >= n:
            p = -p if p < 0 else 2 * n - 2 - p
        return p

    xi = np.array([[refl(x + i - 3, w) for i in range(7)] for x in range(w)])
    yi = np.array([[refl(y + i - 3, h) for i in range(7)] for y in range(h)])
    rows = (src.astype(np.int64)[:, xi] * k).sum(-1)
    cols = (rows[yi, :] * k[None, :, None]).sum(1)
    return np.clip((cols + 32768) >> 16, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h", [(40, 30), (9, 12), (7, 7)])
def test_blur_vs_model(w, h):
    src = np.random.RandomState(w * h).randint(0, 256, size=(h, w)).astype(np.uint8)
    assert np.array_equal(ob.blur(src), _blur_model(src))


def test_blur_saturates():
    src = np.full((20, 20), 255, np.uint8)
    assert (ob.blur(src) == 255).all()  # 255*257*257 overshoots 255<<16: must saturate
    assert (ob.blur(np.zeros((20, 20), np.uint8)) == 0).all()


# ---- fastAtan2, cos/sin ---------------------------------------------------------------------

def test_fast_atan2_accuracy():
    rs = np.random.RandomState(0)
    for _ in range(2000):
        y, x = rs.randint(-200000, 200001, size=2)
        a = ob.fast_atan2(y, x)
        ref = math.degrees(math.atan2(y, x)) % 360.0
        err = abs(a - ref)
        assert min(err, 360 - err) <= 0.3
    assert ob.fast_atan2(0, 0) == 0.0
    assert ob.fast_atan2(0, 5) == 0.0
    assert abs(ob.fast_atan2(5, 0) - 90) < 1e-4
    assert abs(ob.fast_atan2(0, -5) - 180) < 1e-4
    assert abs(ob.fast_atan2(-5, 0) - 270) < 1e-4


def test_cossin_is_rounded_double():
    for deg in (0.0, 24.808466, 90.0, 179.99, 314.1707, 359.9):
        rad = np.float32(np.float32(deg) * np.float32(math.pi / np.float32(180.0)))
        c, s = ob.cossin(rad)
        assert c == np.float32(math.cos(float(rad))) and s == np.float32(math.sin(float(rad)))


# ---- descriptor --------------------------------------------------------------------------------

def test_descriptor_definition():
    from tests.helpers import pattern_pairs

    rs = np.random.RandomState(3)
    img = rs.randint(0, 256, size=(64, 64)).astype(np.uint8)
    pairs = pattern_pairs()
    for ang in (0.0, 37.25, 123.0, 270.0, 359.5):
        d = ob.descriptor(img, 32, 32, ang)
        rad = np.float32(np.float32(ang) * np.float32(np.float32(math.pi) / np.float32(180.0)))
        a, b = np.float32(math.cos(float(rad))), np.float32(math.sin(float(rad)))
        bits = []
        for (x0, y0, x1, y1) in pairs:
            def val(x, y):
                r = int(np.rint(np.float64(np.float32(np.float32(x) * b) + np.float32(np.float32(y) * a))))
                c = int(np.rint(np.float64(np.float32(np.float32(x) * a) - np.float32(np.float32(y) * b))))
                return int(img[32 + r, 32 + c])
            bits.append(1 if val(x0, y0) < val(x1, y1) else 0)
        want = np.packbits(np.array(bits, np.uint8).reshape(32, 8)[:, ::-1], axis=1).reshape(32)
        assert np.array_equal(d, want)


# ---- Hamming -------------------------------------------------------------------------------------

def test_hamming_vs_popcount():
    rs = np.random.RandomState(5)
    q = rs.randint(0, 256, size=(37, 32)).astype(np.uint8)
    t = rs.randint(0, 256, size=(53, 32)).astype(np.uint8)
    t[7] = q[3]
    t[20] = t[10]  # duplicate: first index must win
    want = np.unpackbits(q[:, None, :] ^ t[None, :, :], axis=-1).sum(-1)
    assert np.array_equal(ob.matrix(q, t), want.astype(np.uint16))
    assert ob.distance(q[0], t[0]) == want[0, 0]
    idx, best, second = ob.best2(q, t)
    for i in range(len(q)):
        order = np.argsort(want[i], kind="stable")
        assert idx[i] == order[0] and best[i] == want[i, order[0]] and second[i] == want[i, order[1]]
    assert best[3] == 0 and idx[3] == 7
    i0, b0, s0 = ob.best2(q, t[:0])
    assert (i0 == -1).all() and (b0 == 256).all() and (s0 == 256).all()


# ---- octree ------------------------------------------------------------------------------------

def test_octree_small_cases():
    # fewer keys than the target: every key survives, singletons never split
    keys = np.array([[10, 10, 5], [400, 40, 9], [300, 300, 7]], np.float32)
    out = ob.octree(keys, 16, 624, 16, 464, 10)
    assert sorted(map(tuple, out)) == sorted(map(tuple, keys))
    # reference quirk (S/ORBextractor.cc:682): the loop stops when the NODE COUNT did not change,
    # so a root whose keys all fall in one child ends after one split with a single survivor
    keys = np.array([[10, 10, 5], [100, 40, 9], [300, 200, 7]], np.float32)
    assert [tuple(r) for r in ob.octree(keys, 16, 624, 16, 464, 10)] == [(100, 40, 9)]
    # two keys in one final cell: larger response wins; equal response: first in input order wins
    keys = np.array([[10, 10, 5], [11, 10, 9]], np.float32)
    assert [tuple(r) for r in ob.octree(keys, 16, 624, 16, 464, 1)] == [(11, 10, 9)]
    keys = np.array([[10, 10, 9], [11, 10, 9]], np.float32)
    assert [tuple(r) for r in ob.octree(keys, 16, 624, 16, 464, 1)] == [(10, 10, 9)]


@pytest.mark.parametrize("seed,n,target", [(0, 3000, 217), (1, 500, 60), (2, 150, 151), (3, 5000, 434)])
def test_octree_properties(seed, n, target):
    rs = np.random.RandomState(seed)
    W, H = 608, 448
    pos = rs.permutation(W * H)[:n]
    keys = np.stack([pos % W, pos // W, rs.randint(7, 200, n)], 1).astype(np.float32)
    out = ob.octree(keys, 16, 16 + W, 16, 16 + H, target)
    assert min(n, target) <= len(out) <= target + 3
    inp = set(map(tuple, keys))
    assert all(tuple(r) in inp for r in out) and len(set(map(tuple, out))) == len(out)


# ---- full extractor invariants --------------------------------------------------------------------

def test_extract_invariants():
    o = ob.OrbOracle(1000)
    img = synth.synth_frame(640, 480, 0)
    kps, desc = o.extract(img)
    per = o.tables()["per_level"]
    scale = o.tables()["scale"]
    assert len(kps) == len(desc) and (np.diff(kps["octave"]) >= 0).all()  # level-major
    for l in range(8):
        k = kps[kps["octave"] == l]
        lk = o.level_kps(l)
        assert len(k) == len(lk) and per[l] <= len(k) <= per[l] + 3
        w, h = o.level_size(l)
        assert (lk["x"] >= 19).all() and (lk["x"] < w - 19).all() and (lk["y"] >= 19).all() and (lk["y"] < h - 19).all()
        if l:
            assert np.array_equal(k["x"], lk["x"] * scale[l]) and np.array_equal(k["y"], lk["y"] * scale[l])
        assert (k["size"] == float(int(31 * scale[l]))).all()
        assert (k["angle"] >= 0).all() and (k["angle"] <= 360).all() and (k["class_id"] == -1).all()
        # candidates in (cell row, cell col, y, x) order are unique positions
        c = o.candidates(l)
        assert len(set(map(tuple, c[:, :2]))) == len(c) and (c[:, 2] >= 7).all()
    # padded planes: REFLECT_101 border
    p = o.level(0)
    assert np.array_equal(p[19:-19, 19:-19], img)
    assert np.array_equal(p[19:-19, 18], img[:, 1]) and np.array_equal(p[0, 19:-19], img[19])
    assert np.array_equal(p[-1, 19:-19], img[-20]) and p[0, 0] == img[19, 19]


def test_extract_flat_and_empty():
    o = ob.OrbOracle(1000)
    kps, desc = o.extract(synth.flat_frame(640, 480))
    assert len(kps) == 0 and desc.shape == (0, 32)
    kps, desc = o.extract(np.zeros((0, 0), np.uint8))
    assert len(kps) == 0


def test_extract_fallback_cells_used():
    """A low-contrast frame has cells with no FAST-20 corner: the FAST-7 retry must fire
    (S/ORBextractor.cc:829-833), visible as candidates with 7 <= score < 20."""
    img = (synth.synth_frame(640, 480, 1).astype(np.int32) - 128) // 6 + 128
    o = ob.OrbOracle(1000)
    o.extract(img.astype(np.uint8))
    c = o.candidates(0)
    assert ((c[:, 2] >= 7) & (c[:, 2] < 20)).any() and (c[:, 2] >= 20).any()


def test_distinctive_descriptor_definition():
    """MapPoint::ComputeDistinctiveDescriptors: least median (index floor(0.5*(N-1)) of the sorted row,
    self distance included), first row wins ties."""
    rs = np.random.RandomState(8)
    desc, offs = [], [0]
    for n in (1, 2, 3, 7, 20, 0, 51, 4):
        d = rs.randint(0, 256, (n, 32)).astype(np.uint8)
        if n == 4:
            d[:] = d[0]      # all identical: every median 0, row 0 must win
        desc.append(d)
        offs.append(offs[-1] + n)
    desc = np.concatenate(desc)
    idx, med = ob.distinctive(desc, offs)
    for p in range(len(offs) - 1):
        rows = desc[offs[p]:offs[p + 1]]
        n = len(rows)
        if n == 0:
            assert idx[p] == -1
            continue
        D = np.unpackbits(rows[:, None, :] ^ rows[None, :, :], axis=-1).sum(-1)
        meds = np.sort(D, axis=1)[:, int(0.5 * (n - 1))]
        assert med[p] == meds.min() and idx[p] == int(np.argmin(meds))
