"""Guided search: single-frame host API vs the CPU oracle, and the batched device path (64 frames)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from oracle import bindings as ob  # noqa: E402
from weiner_slamit_v2_amd import api, synth  # noqa: E402
from test_gpu_orb import _search_batch_tensors  # noqa: E402

for n, m in ((1000, 1000), (2000, 3000)):
    frame, q = synth.synth_search(n, m, 3)
    api.ORBmatcher.guided_search(frame, q)
    t0 = time.perf_counter()
    for _ in range(10):
        api.ORBmatcher.guided_search(frame, q)
    tg = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        ob.guided_search(frame, q)
    tc = (time.perf_counter() - t0) / 10
    problems = [synth.synth_search(n, m, 100 + i) for i in range(64)]
    bounds = tuple(problems[0][0][k] for k in ("min_x", "min_y", "inv_w", "inv_h"))
    d = _search_batch_tensors(problems, 2048, 3072)
    s = torch.cuda.Stream()
    api.ORBmatcher.guided_search_batch_dev(d, bounds, stream=s.cuda_stream)
    s.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(5):
        api.ORBmatcher.guided_search_batch_dev(d, bounds, stream=s.cuda_stream)
    e1.record(s)
    s.synchronize()
    tb = e0.elapsed_time(e1) / 5
    print("n=%d m=%d: host API %.3f ms/frame  cpu oracle %.3f ms/frame  batch of 64 on device %.3f ms = %.1f us/frame (%.0fx the CPU)" % (
        n, m, 1e3 * tg, 1e3 * tc, tb, 1e3 * tb / 64, 1e3 * tc / (tb / 64)))
