import sys, time, os
sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, synth
prob = synth.synth_ba(50, 2000, 8)
opt = api.Optimizer(64, 2048, len(prob["edge_kf"]) + 64, 1, 0)
opt.LocalBundleAdjustment(prob)
ts = []
for _ in range(10):
    t0 = time.perf_counter(); out = opt.LocalBundleAdjustment(prob); ts.append(time.perf_counter() - t0)
print(os.environ.get("SLAMIT_BA_NO_BLOCKS"), "ms/window min %.3f med %.3f" % (1e3 * min(ts), 1e3 * sorted(ts)[5]), out["stats"]["n_its"], [sum(t) for t in out["stats"]["trials"]])
opt.profile(True); opt.LocalBundleAdjustment(prob); print(opt.profile_read())
