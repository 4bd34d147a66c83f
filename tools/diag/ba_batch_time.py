"""Where a batch of 64 local-BA windows spends its time end to end (GPU box): Python wrapper vs the C call, and inside the C call
(SLAMIT_BA_TIMING=1 prints the host phases of slamit_ba_solve_batch).  tools/diag/ba_batch_time.py [window8|dense]"""
import sys, time
sys.path.insert(0, ".")
import ctypes as C
from weiner_slamit_v2_amd import api, synth
obs = None if (len(sys.argv) > 1 and sys.argv[1] == "dense") else 8
probs = [synth.synth_ba(50, 2000, obs, seed=12345 + i) for i in range(64)]
ne = max(len(q["edge_kf"]) for q in probs)
opt = api.Optimizer(64, 2048, ne + 64, 64, 0)
opt.LocalBundleAdjustmentBatch(probs)
for rep in range(3):
    t0 = time.perf_counter()
    n = len(probs)
    P = (api.BaProblem * n)(); R = (api.BaResult * n)()
    keeps, outs, sts = [], [], []
    for i, prob in enumerate(probs):
        P[i], keep = api._ba_problem(prob); keeps.append(keep)
        R[i], out, st = opt._result(P[i].n_kf, P[i].n_pt, P[i].n_edge); outs.append(out); sts.append(st)
    t1 = time.perf_counter()
    o = api.BaOpts(5, 10, api.HUBER_MONO, 5.991, None)
    api._check(api.lib().slamit_ba_solve_batch(opt._h, n, P, C.byref(o), R), "solve")
    t2 = time.perf_counter()
    its = sum(sum(st.n_its) for st in sts)
    print("python prep %.2f ms, C call %.2f ms, total %.2f ms -> %.0f it/s (C call alone %.0f it/s)" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t2 - t0), its / (t2 - t0), its / (t2 - t1)))
