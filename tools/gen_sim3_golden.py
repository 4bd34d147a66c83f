#!/usr/bin/env python3
"""Generates tests/golden/sim3_*.npz from the REFERENCE's own g2o (oracle/_ref/libba_ref.so: sim3_ref_solve =
Optimizer::OptimizeSim3's graph and schedule on POD inputs).  Authoring container only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bindings as ob  # noqa: E402
from weiner_slamit_v2_amd import synth  # noqa: E402

CASES = {  # name: (n, outlier fraction, seed, initial perturbation, fix_scale)
    "typical": (200, 0.15, 0, 0.03, False),
    "many_outliers": (600, 0.50, 5, 0.03, False),
    "fixed_scale": (100, 0.10, 2, 0.03, True),     # bFixScale (stereo / RGB-D loop closing): column 6 of every Jacobian is zero
    "clean": (60, 0.0, 7, 0.02, False),            # nothing dropped: 5 more iterations instead of 10 (Optimizer.cc:1206-1210)
    "twelve": (12, 0.0, 3, 0.03, False),
    "under10": (9, 0.0, 4, 0.03, False),           # < 10 pairs left: returns 0, g2oS12 untouched (:1212-1213)
    "rough": (300, 0.25, 8, 0.12, False),          # far initial estimate: rejected LM trials
}
KEYS = ("p1", "p2", "obs1", "obs2", "inv_sigma2_1", "inv_sigma2_2", "intr1", "intr2", "r12", "t12")


def main():
    assert ob.ba_ref_available()
    for name, (n, of, seed, pert, fix) in CASES.items():
        pr = synth.synth_sim3(n, of, seed, pert, fix)
        r = ob.sim3_ref_solve(pr)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sim3_%s.npz" % name),
                            **{k: np.asarray(pr[k], np.float64) for k in KEYS}, s12=pr["s12"], th2=pr["th2"], fix_scale=pr["fix_scale"],
                            ref_r12=r["r12"], ref_t12=r["t12"], ref_s12=r["s12"], ref_inlier=r["inlier"], ref_n_inliers=r["n_inliers"],
                            ref_n_its=np.array(r["n_its"]), ref_chi2=np.array(r["chi2"]))
        print(name, n, "inliers", r["n_inliers"], "its", r["n_its"], "chi2", [round(c, 3) for c in r["chi2"]])


if __name__ == "__main__":
    main()
