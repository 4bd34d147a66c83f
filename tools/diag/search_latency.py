"""Single-frame latency of slamit_guided_search (host pointers): the conflict-heavy synthetic case of the tests and a
tracking-like one (every map point aims at its own keypoint)."""
import sys, time
sys.path.insert(0, ".")
from oracle import bindings as ob  # noqa: E402
from weiner_slamit_v2_amd import api, synth  # noqa: E402

for name, kw in (("half of the queries re-target a taken keypoint", dict(retarget=True)), ("distinct targets (tracking-like)", dict(retarget=False))):
    frame, q = synth.synth_search(1000, 1000, 3, **kw)
    api.ORBmatcher.guided_search(frame, q)
    r = []
    for _ in range(50):
        t0 = time.perf_counter(); g = api.ORBmatcher.guided_search(frame, q); r.append(time.perf_counter() - t0)
    r.sort()
    t0 = time.perf_counter()
    for _ in range(10): ob.guided_search(frame, q)
    tc = (time.perf_counter() - t0) / 10
    print("%-50s HIP median %.3f ms  CPU oracle %.3f ms  (%d matches)" % (name, 1e3 * r[25], 1e3 * tc, g[1]))
