"""Aggregate throughput of single-window local-BA solves on N concurrent worker handles (what the pipeline config does per rank at 8 GPUs):
python3 tools/diag/ba_workers_time.py [windows per batch]"""
import sys
import time

sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, shard, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
probs = [synth.synth_ba(50, 2000, 8, seed=12345 + i) for i in range(B)]
ne = max(len(p["edge_kf"]) for p in probs) + 64
for n in (1, 2, 3, 4, 6, 8):
    w = shard.BaWorkers(lambda: api.Optimizer(64, 2048, ne, B, 0), n)
    for j in [w.submit(probs) for _ in range(n)]:
        w.result(j)
    t0 = time.perf_counter()
    jobs = [w.submit(probs) for _ in range(60)]
    for j in jobs:
        w.result(j)
    el = time.perf_counter() - t0
    print("%d worker(s), %d window(s) per batch: %.3f ms per batch, %.0f windows/s" % (n, B, 1e3 * el / 60, 60 * B / el), flush=True)
    w.close()
