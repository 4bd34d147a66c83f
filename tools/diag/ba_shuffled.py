"""GPU box: a window-8 window (50 x 2000) with its keyframes along the trajectory, in a random order, and in that order without the
column renumbering (SLAMIT_BA_KEEP_ORDER=1): ms per solve.  tools/diag/ba_shuffled.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, synth

base = synth.synth_ba(50, 2000, 8)
n = len(base["kf_fixed"])
perm = np.concatenate([[0], 1 + np.random.RandomState(5).permutation(n - 1)])
inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
shuf = dict(base)
for k in ("kf_pose", "kf_fixed", "kf_intr"):
    shuf[k] = np.ascontiguousarray(np.asarray(base[k])[perm])
shuf["edge_kf"] = inv[np.asarray(base["edge_kf"])].astype(np.int32)
opt = api.Optimizer(64, 2048, len(base["edge_kf"]) + 64, 1, 0)


def run(tag, prob):
    opt.LocalBundleAdjustment(prob)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); out = opt.LocalBundleAdjustment(prob); ts.append(time.perf_counter() - t0)
    print("%-34s %.3f ms  its %s" % (tag, 1e3 * sorted(ts)[5], out["stats"]["n_its"]))


run("along the trajectory", base)
run("shuffled (renumbered)", shuf)
os.environ["SLAMIT_BA_KEEP_ORDER"] = "1"
run("shuffled, caller's order", shuf)
