"""One step's B streams as G groups of B / G on G HIP streams (own extractor handle each) against one group: python3 tools/diag/two_halves.py [B] [G ...]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
groups = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
dev = torch.device("cuda", 0)
uniq = 16
fa = [synth.synth_frame(640, 480, i) for i in range(uniq)]
fb = [synth.warp_frame(fa[i], i) for i in range(uniq)]
for G in groups + groups:
    b = B // G
    frames = [torch.from_numpy(np.stack([f[i % uniq] for i in range(b)])).to(dev) for f in (fa, fb)]
    exts, bufs, streams = [], [], []
    for g in range(G):
        e = api.ORBextractor(1000, 1.2, 8, 20, 7, device=0, max_batch=b)
        e._bind(640, 480, b)
        cap = e.max_keypoints
        exts.append(e)
        bufs.append(dict(kps=[torch.zeros((b, cap, 7), dtype=torch.float32, device=dev) for _ in range(3)],
                         desc=[torch.zeros((b, cap, 32), dtype=torch.uint8, device=dev) for _ in range(3)],
                         n=[torch.zeros(b, dtype=torch.int32, device=dev) for _ in range(3)],
                         idx=torch.zeros((b, cap), dtype=torch.int32, device=dev), best=torch.zeros((b, cap), dtype=torch.int32, device=dev),
                         second=torch.zeros((b, cap), dtype=torch.int32, device=dev)))
        streams.append(torch.cuda.Stream(dev))

    def step(k):
        cur, prv = k % 3, (k - 1) % 3
        for g in range(G):
            s = streams[g].cuda_stream
            d = bufs[g]
            exts[g].extract_batch_dev(frames[k & 1], d["kps"][cur], d["desc"][cur], d["n"][cur], stream=s)
            api.ORBmatcher.best2_batch_dev(d["desc"][cur], d["n"][cur], d["desc"][prv], d["n"][prv], d["idx"], d["best"], d["second"], cap, device=0, stream=s)

    for k in range(4):
        step(k)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for k in range(4 + 20 * rep, 24 + 20 * rep):
            step(k)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 20)
    ms = 1e3 * sorted(ts)[2]
    print("B %d in %d group(s): %.4f ms per step, %.0f frames/s" % (B, G, ms, B / ms * 1e3), flush=True)
    del exts, bufs
