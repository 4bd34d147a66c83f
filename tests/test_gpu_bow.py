"""GPU parity: the vocabulary-node search (slamit_bow_search) vs the CPU oracle's sequential restatement of the BoW drivers'
loops (ORBmatcher.cc:161-290, 526-657 SearchByBoW; 659-826 SearchForTriangulation).  Bar: identical match and distance
arrays (integer work; mode 1's float gates are written like the reference's and compiled without contraction)."""
import numpy as np
import pytest

from oracle import bindings as ob
from weiner_slamit_v2_amd import api, synth

pytestmark = pytest.mark.gpu


def _both(s1, s2, g, **kw):
    a = api.ORBmatcher.bow_search(s1, s2, g, **kw)
    o = ob.bow_search(s1, s2, g, **kw)
    assert np.array_equal(a[0], o[0]), "match12 differs at %s" % np.nonzero(a[0] != o[0])[0][:8]
    assert np.array_equal(a[1], o[1]), "dist12 differs at %s" % np.nonzero(a[1] != o[1])[0][:8]
    assert a[2] == o[2] == int((o[0] >= 0).sum())
    return o


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("inclusive", [True, False])
def test_search_by_bow_matches_oracle(seed, inclusive):
    n1, n2, nodes = [(1000, 1000, 100), (1500, 700, 40), (300, 2000, 7), (1000, 1000, 1000), (64, 64, 1), (2000, 2000, 150)][seed]
    s1, s2, g, _ = synth.synth_bow(n1, n2, nodes, seed, mode=0)
    m, d, nm = _both(s1, s2, g, mode=0, th=50, th_inclusive=inclusive, nnratio=0.6)
    assert nm > 0
    # a candidate is matched by at most one query
    taken = m[m >= 0]
    assert len(np.unique(taken)) == len(taken)


@pytest.mark.parametrize("seed", range(6))
def test_search_for_triangulation_matches_oracle(seed):
    s1, s2, g, epi = synth.synth_bow(1200, 1100, 90, seed, mode=1)
    m, d, nm = _both(s1, s2, g, mode=1, th=50, epi=epi)
    assert (nm > 0) == (seed % 3 != 0)          # seed % 3 == 0 is the degenerate F12 = 0 case: no candidate passes
    if nm:
        assert (d[m >= 0] <= 50).all()


def test_groups_larger_than_a_wavefront_and_ratio_extremes():
    s1, s2, g, _ = synth.synth_bow(400, 1500, 3, 11, mode=0, big_group=700)
    assert (g["c_ptr"][1:] - g["c_ptr"][:-1]).max() > 640
    for nnratio, th in ((0.6, 50), (0.9, 100), (1.5, 256), (0.0, 50)):
        _both(s1, s2, g, mode=0, th=th, th_inclusive=True, nnratio=nnratio)
    s1, s2, g, epi = synth.synth_bow(400, 1500, 3, 13, mode=1, big_group=700)
    _both(s1, s2, g, mode=1, th=60, epi=epi)


def test_empty_and_degenerate_inputs():
    s1, s2, g, _ = synth.synth_bow(50, 60, 5, 2, mode=0)
    empty = dict(q_ptr=np.zeros(1, np.int32), q_idx=np.zeros(0, np.int32), c_ptr=np.zeros(1, np.int32), c_idx=np.zeros(0, np.int32))
    m, d, nm = api.ORBmatcher.bow_search(s1, s2, empty)
    assert nm == 0 and (m == -1).all() and (d == 256).all()
    # groups with queries but no candidates and the other way round
    g2 = dict(q_ptr=np.array([0, 3, 3], np.int32), q_idx=np.array([0, 1, 2], np.int32), c_ptr=np.array([0, 0, 4], np.int32),
              c_idx=np.array([0, 1, 2, 3], np.int32))
    _both(s1, s2, g2, mode=0, th=256, th_inclusive=True, nnratio=2.0)
    # every query / candidate masked out
    s1z = dict(s1, valid=np.zeros(50, np.uint8))
    assert _both(s1z, s2, g)[2] == 0
    s2z = dict(s2, valid=np.zeros(60, np.uint8))
    assert _both(s1, s2z, g)[2] == 0
    # no features at all on one side
    none = dict(desc=np.zeros((0, 32), np.uint8), valid=None)
    assert api.ORBmatcher.bow_search(none, s2, empty)[2] == 0
    m, d, nm = api.ORBmatcher.bow_search(s1, none, empty)
    assert nm == 0 and len(m) == 50


def test_identical_descriptors_first_candidate_wins_and_ratio_rejects():
    d = np.random.RandomState(3).randint(0, 256, (1, 32)).astype(np.uint8)
    s1 = dict(desc=np.repeat(d, 2, 0), valid=None)
    s2 = dict(desc=np.repeat(d, 3, 0), valid=None)
    g = dict(q_ptr=np.array([0, 2], np.int32), q_idx=np.array([0, 1], np.int32), c_ptr=np.array([0, 3], np.int32), c_idx=np.array([2, 0, 1], np.int32))
    # best = second = 0: 0 < nnratio * 0 is false -> nothing accepted, whatever the ratio
    m, dd, nm = _both(s1, s2, g, mode=0, th=50, th_inclusive=True, nnratio=0.9)
    assert nm == 0 and (dd == 0).all()
    # with ONE candidate the second distance is 256: accepted, and the second query finds its candidate gone
    g1 = dict(g, c_ptr=np.array([0, 1], np.int32), c_idx=np.array([2], np.int32))
    m, dd, nm = _both(s1, s2, g1, mode=0, th=50, th_inclusive=True, nnratio=0.9)
    assert list(m) == [2, -1] and list(dd) == [0, 256] and nm == 1
    # exclusive threshold: best == th is rejected, inclusive accepts
    s2b = dict(desc=d.copy(), valid=None)
    s2b["desc"][0, :6] ^= 0xFF                                   # 48 bits
    s2b["desc"][0, 6] ^= 0x03                                    # 50 bits
    g1b = dict(q_ptr=np.array([0, 1], np.int32), q_idx=np.array([0], np.int32), c_ptr=np.array([0, 1], np.int32), c_idx=np.array([0], np.int32))
    assert _both(s1, s2b, g1b, mode=0, th=50, th_inclusive=True, nnratio=0.9)[2] == 1
    assert _both(s1, s2b, g1b, mode=0, th=50, th_inclusive=False, nnratio=0.9)[2] == 0


def test_argument_errors_are_reported():
    s1, s2, g, _ = synth.synth_bow(50, 60, 5, 4, mode=0)
    bad = dict(g, c_idx=g["c_idx"].copy())
    bad["c_idx"][1] = bad["c_idx"][0]                            # a side-2 feature in two places
    with pytest.raises(RuntimeError, match="repeated"):
        api.ORBmatcher.bow_search(s1, s2, bad)
    bad = dict(g, q_idx=g["q_idx"].copy())
    bad["q_idx"][0] = 50
    with pytest.raises(RuntimeError, match="out of range"):
        api.ORBmatcher.bow_search(s1, s2, bad)
    n2 = 2048 + 1                                                # SLAMIT_BOW_MAX_GROUP + 1
    big1 = dict(desc=np.zeros((1, 32), np.uint8), valid=None)
    big2 = dict(desc=np.zeros((n2, 32), np.uint8), valid=None)
    gb = dict(q_ptr=np.array([0, 1], np.int32), q_idx=np.array([0], np.int32), c_ptr=np.array([0, n2], np.int32), c_idx=np.arange(n2, dtype=np.int32))
    with pytest.raises(RuntimeError, match="SLAMIT_BOW_MAX_GROUP"):
        api.ORBmatcher.bow_search(big1, big2, gb)
    with pytest.raises(RuntimeError, match="mode 1 needs"):
        _mode1_without_geometry(s1, s2, g)


def _mode1_without_geometry(s1, s2, g):
    import ctypes as C
    rule = api.BowRule()
    rule.mode, rule.th = 1, 50
    qp, qi, cp, ci = (np.ascontiguousarray(g[k], np.int32) for k in ("q_ptr", "q_idx", "c_ptr", "c_idx"))
    gg = api.BowGroups(len(qp) - 1, qp.ctypes.data, qi.ctypes.data, cp.ctypes.data, ci.ctypes.data)
    m = np.zeros(50, np.int32)
    nm = C.c_int32(0)
    d1, d2 = np.ascontiguousarray(s1["desc"]), np.ascontiguousarray(s2["desc"])
    api._check(api.lib().slamit_bow_search(0, d1.ctypes.data_as(C.c_void_p), 50, None, d2.ctypes.data_as(C.c_void_p), 60, None,
                                           C.byref(gg), C.byref(rule), m.ctypes.data_as(C.c_void_p), None, C.byref(nm)), "slamit_bow_search")
