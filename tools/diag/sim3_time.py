"""Times slamit_sim3_optimize_batch (host-pointer C-ABI) next to the CPU oracle and, when present, the reference's g2o."""
import sys, time
sys.path.insert(0, ".")
from oracle import bindings as ob  # noqa: E402
from weiner_slamit_v2_amd import api, synth  # noqa: E402

probs = [synth.synth_sim3(300, 0.2, 200 + i, 0.03) for i in range(64)]
api.Optimizer.OptimizeSim3(probs[:2])
for nb in (1, 64):
    res = []
    for _ in range(5):
        t0 = time.perf_counter(); api.Optimizer.OptimizeSim3(probs[:nb]); res.append(time.perf_counter() - t0)
    print("HIP batch of %2d problems (300 pairs): %.3f ms total, %.3f ms per problem" % (nb, 1e3 * min(res), 1e3 * min(res) / nb))
t0 = time.perf_counter()
for p in probs[:8]: ob.sim3_solve(p)
print("CPU oracle: %.3f ms per problem" % (1e3 * (time.perf_counter() - t0) / 8))
if ob.ba_ref_available():
    t0 = time.perf_counter()
    for p in probs[:8]: ob.sim3_ref_solve(p)
    print("reference g2o: %.3f ms per problem" % (1e3 * (time.perf_counter() - t0) / 8))
