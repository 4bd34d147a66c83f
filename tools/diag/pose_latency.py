"""Single-call latency of the per-frame optimisers through the host-pointer C-ABI."""
import sys, time
sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, synth  # noqa: E402

pp = synth.synth_pose(500, 0.15, 3)
ps = synth.synth_sim3(300, 0.2, 3)
api.Optimizer.PoseOptimization(pp); api.Optimizer.OptimizeSim3(ps)
for name, fn, arg in (("PoseOptimization (500 obs)", api.Optimizer.PoseOptimization, pp), ("OptimizeSim3 (300 pairs)", api.Optimizer.OptimizeSim3, ps)):
    res = []
    for _ in range(30):
        t0 = time.perf_counter(); fn(arg); res.append(time.perf_counter() - t0)
    res.sort()
    print("%-28s median %.3f ms  min %.3f ms" % (name, 1e3 * res[15], 1e3 * res[0]))
