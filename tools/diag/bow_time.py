"""Times slamit_bow_search (host-pointer C-ABI: upload + node loops + download) next to the CPU oracle's sequential loops."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from oracle import bindings as ob  # noqa: E402
from weiner_slamit_v2_amd import api, synth  # noqa: E402

for name, (n1, n2, nodes), mode in (("SearchByBoW 1000x1000, 100 nodes", (1000, 1000, 100), 0), ("SearchByBoW 2000x2000, 100 nodes", (2000, 2000, 100), 0),
                                    ("SearchForTriangulation 1000x1000, 100 nodes", (1000, 1000, 100), 1), ("SearchByBoW 2000x2000, 10 nodes", (2000, 2000, 10), 0)):
    s1, s2, g, epi = synth.synth_bow(n1, n2, nodes, 1, mode=mode)
    kw = dict(mode=mode, th=50, th_inclusive=True, nnratio=0.6, epi=epi)
    api.ORBmatcher.bow_search(s1, s2, g, **kw)
    t0 = time.perf_counter()
    for _ in range(50):
        a = api.ORBmatcher.bow_search(s1, s2, g, **kw)
    tg = (time.perf_counter() - t0) / 50
    t0 = time.perf_counter()
    for _ in range(20):
        o = ob.bow_search(s1, s2, g, **kw)
    tc = (time.perf_counter() - t0) / 20
    assert np.array_equal(a[0], o[0])
    print("%-48s HIP %.3f ms  CPU oracle %.3f ms  (%d matches)" % (name, 1e3 * tg, 1e3 * tc, a[2]))
