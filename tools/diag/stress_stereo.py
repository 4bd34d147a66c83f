"""Randomised parity stress of the stereo paths: random BA windows and pose problems with stereo observations against the CPU oracle.
Reports how often the LM path (iteration counts) differs and the distribution of the state differences: the reference's float inverse depth
(cam_project) makes the error function discontinuous at the 1e-13 level, so last-bit differences can be amplified near convergence."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle import bindings as ob  # noqa: E402
from weiner_slamit_v2_amd import api, synth  # noqa: E402

t_end = time.time() + (float(sys.argv[1]) if len(sys.argv) > 1 else 120)
rs = np.random.RandomState(4242)
opt = api.Optimizer(64, 2048, 110000, 1, 0)
n_ba = n_pose = its_diff = flag_diff = pose_bad = 0
d_all, d_mixed, dp_pose = [], [], []
while time.time() < t_end:
    seed = int(rs.randint(0, 1 << 30))
    nk, npt = int(rs.randint(3, 51)), int(rs.randint(20, 1200))
    obs = None if rs.rand() < 0.2 else int(rs.randint(2, min(nk, 12) + 1))
    sf = float(rs.choice([1.0, 0.8, 0.5, 0.2]))
    prob = synth.synth_ba(nk, npt, obs, outlier_frac=float(rs.choice([0.0, 0.03, 0.1])), seed=seed, n_fixed=int(rs.randint(1, min(nk, 4))), stereo_frac=sf)
    g, o = opt.LocalBundleAdjustment(prob), ob.ba_solve(prob)
    n_ba += 1
    its_diff += list(g["stats"]["n_its"]) != list(o["stats"]["n_its"]) or g["stats"]["trials"] != o["stats"]["trials"]
    flag_diff += not np.array_equal(g["edge_outlier"], o["edge_outlier"])
    d = max(np.abs(g["kf_pose"] - o["kf_pose"]).max() / max(np.abs(o["kf_pose"]).max(), 1.0), np.abs(g["pt_xyz"] - o["pt_xyz"]).max() / max(np.abs(o["pt_xyz"]).max(), 1.0))
    (d_all if sf == 1.0 else d_mixed).append(d)
    pp = synth.synth_pose(int(rs.randint(3, 1500)), float(rs.choice([0.0, 0.15, 0.4])), seed % 100000, float(rs.choice([0.01, 0.05])), stereo_frac=float(rs.choice([1.0, 0.5, 0.2])))
    gp, op = api.Optimizer.PoseOptimization([pp])[0], ob.pose_solve(pp)
    n_pose += 1
    dpp = np.abs(gp["pose"] - op["pose"]).max() / max(np.abs(op["pose"]).max(), 1.0)
    dp_pose.append(dpp)
    pose_bad += gp["n_inliers"] != op["n_inliers"] or not np.array_equal(gp["outlier"], op["outlier"]) or dpp > 1e-5
q = lambda a: "n %d median %.1e p90 %.1e p99 %.1e max %.1e" % (len(a), np.median(a), np.percentile(a, 90), np.percentile(a, 99), np.max(a)) if len(a) else "n 0"
print("BA windows %d: LM path differs on %d, final outlier flags differ on %d" % (n_ba, its_diff, flag_diff))
print("  all-stereo windows, relative state difference: " + q(d_all))
print("  mixed windows, relative state difference:      " + q(d_mixed))
print("pose problems %d: failures (inlier set / 1e-5) %d; relative pose difference: %s" % (n_pose, pose_bad, q(dp_pose)))
