"""Times slamit_pose_optimize_batch: 256 frames x 500 observations (host-pointer C-ABI, upload + solve + download)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, synth  # noqa: E402

probs = [synth.synth_pose(500, seed=i) for i in range(256)]
api.Optimizer.PoseOptimization(probs)
res = []
for _ in range(5):
    t0 = time.perf_counter()
    api.Optimizer.PoseOptimization(probs)
    res.append(time.perf_counter() - t0)
print("batch of 256 frames: %.2f ms  %.0f frames/s" % (1e3 * min(res), 256 / min(res)))
