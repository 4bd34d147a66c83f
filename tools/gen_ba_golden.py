#!/usr/bin/env python3
"""Generates tests/golden/ba_*.npz from the REFERENCE's own g2o (oracle/_ref/libba_ref.so, built by
oracle/Makefile.ref from /root/reference).  Authoring container only.  Each file holds the POD
inputs of one local-BA window (as include/slamit.h lays them out) and what the reference's g2o
produced for it through the Optimizer.cc:507-743 schedule: final poses / points, per-edge chi2,
stage-1 and final outlier flags, per-iteration robust cost, lambda and LM trial counts.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bindings as ob  # noqa: E402
from weiner_slamit_v2_amd import synth  # noqa: E402

CASES = {
    # name: (n_kf, n_pt, obs_per_pt, seed, n_fixed, pose_sigma_scale)
    "tiny": (5, 50, 3, 1, 1, 1.0),
    "small": (10, 200, 4, 2, 1, 1.0),
    "fixed3": (12, 300, 5, 7, 3, 1.0),
    "rough": (8, 150, 4, 11, 1, 12.0),      # large initial error: rejected LM trials
    "rejects": (8, 150, 4, 13, 1, 60.0),    # very large initial error: LM iterations with 3-4 rejected trials
    "allout": (8, 150, 4, 11, 1, 200.0),    # hopeless start: every edge fails the gate, stage 2 has nothing to do
    "window8": (50, 2000, 8, 12345, 1, 1.0),  # BASELINE config 4 geometry, 16,000 edges
    # Optimizer::BundleAdjustment / GlobalBundleAdjustemnt (Optimizer.cc:40-238): ONE optimize(nIterations), Huber delta
    # sqrt(5.99), keyframe 0 fixed -- the same solve with schedule (nIterations, 0)
    "global_init": (2, 150, None, 21, 1, 1.0),   # Tracking::CreateInitialMapMonocular: 2 keyframes, GlobalBundleAdjustemnt(mpMap, 20)
    "global_map": (14, 500, 5, 22, 1, 4.0),      # a small map, 10 iterations
    # windows with stereo observations (EdgeStereoSE3ProjectXYZ, Optimizer.cc:621-650): a 7th field = the fraction of stereo edges
    "stereo_all": (10, 200, 4, 51, 1, 1.0, 1.0),        # every observation stereo (an RGB-D session)
    "stereo_mixed": (20, 600, 6, 52, 2, 1.0, 0.5),      # half of the observations without a right-image match
    "stereo_rough": (8, 150, 4, 56, 1, 80.0, 0.6),      # very large initial error: a rejected LM trial, most edges gated out
    "stereo_window8": (50, 2000, 8, 54, 1, 1.0, 0.8),   # BASELINE config 4 geometry with 80 % stereo observations
    # a window-8 window whose keyframes are listed in a random order (ORB-SLAM2 lists them by co-visibility weight, Optimizer.cc:456-470):
    # the free keyframes are renumbered on the host (csrc/ba_api.hip ba_order_columns) -- an 8th field = the seed of the permutation
    "shuffled": (50, 1000, 8, 61, 2, 1.0, 0.0, 9),
}
# name -> (its_robust, its_final, huber_delta); everything else uses the local-BA schedule of Optimizer.cc:507-743
SCHEDULE = {"global_init": (20, 0, float(np.float32(np.sqrt(5.99)))), "global_map": (10, 0, float(np.float32(np.sqrt(5.99))))}


def make(name):
    k, p, o, seed, nfix, rough = CASES[name][:6]
    prob = synth.synth_ba(k, p, o, seed=seed, n_fixed=nfix, stereo_frac=CASES[name][6] if len(CASES[name]) > 6 else 0.0)
    if rough != 1.0:
        rs = np.random.RandomState(seed + 99)
        prob["pt_xyz"] = (prob["pt_xyz"] + rs.normal(0, 0.02 * rough, prob["pt_xyz"].shape)).astype(np.float32).astype(np.float64)
        for i in range(nfix, k):
            dR, dt = synth.se3_exp(rs.normal(0, 0.005 * rough, 6))
            R = prob["kf_pose"][i, :9].reshape(3, 3)
            t = prob["kf_pose"][i, 9:]
            prob["kf_pose"][i, :9] = (dR @ R).reshape(-1)
            prob["kf_pose"][i, 9:] = dR @ t + dt
        prob["kf_pose"] = prob["kf_pose"].astype(np.float32).astype(np.float64)
    if len(CASES[name]) > 7:   # the same window, its keyframes in a random order
        perm = np.random.RandomState(CASES[name][7]).permutation(k)   # new position i holds old keyframe perm[i]
        inv = np.empty(k, np.int64)
        inv[perm] = np.arange(k)
        for key in ("kf_pose", "kf_fixed", "kf_intr"):
            prob[key] = np.ascontiguousarray(np.asarray(prob[key])[perm])
        prob["edge_kf"] = inv[np.asarray(prob["edge_kf"])].astype(np.int32)
    return prob


def main():
    assert ob.ba_ref_available(), "build oracle/_ref first: make -C oracle -f Makefile.ref"
    out_dir = os.path.join(ROOT, "tests", "golden")
    only = sys.argv[1:]   # optional: the case names to (re)generate; default all
    for name in CASES:
        if only and not any(name.startswith(o) for o in only):
            continue
        prob = make(name)
        sched = SCHEDULE.get(name)
        ref = ob.ba_ref_solve(prob, *sched) if sched else ob.ba_ref_solve(prob)
        st = ref["stats"]
        pad = lambda rows: np.array([list(r) + [np.nan] * (32 - len(r)) for r in rows], dtype=np.float64)
        np.savez_compressed(
            os.path.join(out_dir, "ba_%s.npz" % name),
            kf_pose=prob["kf_pose"].astype(np.float32), kf_fixed=prob["kf_fixed"], kf_intr=prob["kf_intr"].astype(np.float32),
            pt_xyz=prob["pt_xyz"].astype(np.float32), edge_kf=prob["edge_kf"].astype(np.int16), edge_pt=prob["edge_pt"].astype(np.int16),
            edge_uv=prob["edge_uv"].astype(np.float32), edge_inv_sigma2=prob["edge_inv_sigma2"].astype(np.float32),
            **({"edge_ur": prob["edge_ur"].astype(np.float32), "kf_bf": prob["kf_bf"].astype(np.float32)} if "edge_ur" in prob else {}),
            ref_kf_pose=ref["kf_pose"], ref_pt_xyz=ref["pt_xyz"], ref_edge_chi2=ref["edge_chi2"],
            ref_edge_outlier=ref["edge_outlier"], ref_edge_stage1_outlier=ref["edge_stage1_outlier"],
            ref_n_its=np.array(st["n_its"]), ref_chi2=pad(st["chi2"]), ref_lambda=pad(st["lambda"]),
            ref_trials=pad(st["trials"]), ref_chi2_init=np.array(st["chi2_init"]),
            schedule=np.array(sched if sched else (5, 10, ob.HUBER_MONO), np.float64))
        print(name, "edges", len(prob["edge_kf"]), "its", st["n_its"], "trials", st["trials"],
              "outliers", int(ref["edge_outlier"].sum()))


if __name__ == "__main__":
    main()
