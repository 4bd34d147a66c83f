"""Times slamit_hamming_best2_batch_dev alone: 64 pairs of 1000 x 1000 random descriptors."""
import sys

import torch

sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api  # noqa: E402

P, N, cap = 64, 1000, 1024
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randint(0, 256, (P, cap, 32), dtype=torch.uint8, device="cuda", generator=g)
t = torch.randint(0, 256, (P, cap, 32), dtype=torch.uint8, device="cuda", generator=g)
n = torch.full((P,), N, dtype=torch.int32, device="cuda")
idx = torch.zeros((P, cap), dtype=torch.int32, device="cuda")
best, second = torch.zeros_like(idx), torch.zeros_like(idx)
s = torch.cuda.Stream()
torch.cuda.synchronize()
for _ in range(3):
    api.ORBmatcher.best2_batch_dev(q, n, t, n, idx, best, second, cap, device=0, stream=s.cuda_stream)
s.synchronize()
res = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(20):
        api.ORBmatcher.best2_batch_dev(q, n, t, n, idx, best, second, cap, device=0, stream=s.cuda_stream)
    e1.record(s)
    s.synchronize()
    res.append(1e3 * e0.elapsed_time(e1) / 20)
print("us per call: min %.1f  median %.1f" % (min(res), sorted(res)[2]))
