import sys, time
sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, synth
prob = synth.synth_ba(50, 2000, None)
opt = api.Optimizer(64, 2048, len(prob["edge_kf"]) + 64, 1, 0)
out = opt.LocalBundleAdjustment(prob)
t0 = time.perf_counter()
for _ in range(5):
    out = opt.LocalBundleAdjustment(prob)
el = (time.perf_counter() - t0) / 5
print("dense ms/window %.3f  its %s trials %s  -> %.0f LM it/s" % (1e3 * el, out["stats"]["n_its"], [sum(t) for t in out["stats"]["trials"]], sum(out["stats"]["n_its"]) / el))
