"""Per-phase times of an LM trial slot for a batch of local-BA windows (slamit_ba_profile): tools/diag/ba_batch_phases.py [nwin] [dense]"""
import sys
sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, synth
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 64
obs = None if "dense" in sys.argv else 8
probs = [synth.synth_ba(50, 2000, obs, seed=12345 + i) for i in range(nwin)]
ne = max(len(q["edge_kf"]) for q in probs)
opt = api.Optimizer(64, 2048, ne + 64, nwin, 0)
opt.LocalBundleAdjustmentBatch(probs)
opt.profile(True)
outs = opt.LocalBundleAdjustmentBatch(probs)
p = opt.profile_read()
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in p.items()} if isinstance(p, dict) else p)
