import sys, time
sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, synth
prob = synth.synth_ba(50, 2000, 8)
opt = api.Optimizer(64, 2048, len(prob["edge_kf"]) + 64, 1, 0)
opt.LocalBundleAdjustment(prob)
t0 = time.perf_counter()
for _ in range(5):
    opt.LocalBundleAdjustment(prob)
print("ms/window", 1e3 * (time.perf_counter() - t0) / 5)
