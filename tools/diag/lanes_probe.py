#!/usr/bin/env python3
"""GPU box: extract+match throughput when the B camera streams of a step are split into L lanes, each lane with its own
extractor handle, HIP stream and result buffers (the lanes' kernels overlap on the chip).  tools/diag/lanes_probe.py [B] [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from weiner_slamit_v2_amd import api, synth

W, H, NFEAT = (1280, 720, 2000) if os.environ.get("P720") else (640, 480, 1000)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
fa = [synth.synth_frame(W, H, i) for i in range(16)]
fb = [synth.warp_frame(fa[i], i) for i in range(16)]


def run(L, stagger):
    b = B // L
    lanes = []
    for l in range(L):
        fr = [torch.from_numpy(np.stack([f[(l * b + i) % 16] for i in range(b)])).to(dev) for f in (fa, fb)]
        ext = api.ORBextractor(NFEAT, 1.2, 8, 20, 7, device=0, max_batch=b)
        ext._bind(W, H, b)
        cap = ext.max_keypoints
        st = torch.cuda.Stream(dev)
        lanes.append(dict(fr=fr, ext=ext, cap=cap, st=st,
                          kps=[torch.zeros((b, cap, 7), dtype=torch.float32, device=dev) for _ in range(3)],
                          desc=[torch.zeros((b, cap, 32), dtype=torch.uint8, device=dev) for _ in range(3)],
                          n=[torch.zeros(b, dtype=torch.int32, device=dev) for _ in range(3)],
                          idx=torch.zeros((b, cap), dtype=torch.int32, device=dev), best=torch.zeros((b, cap), dtype=torch.int32, device=dev),
                          second=torch.zeros((b, cap), dtype=torch.int32, device=dev)))

    def step(k):
        cur, prv = k % 3, (k - 1) % 3
        for ln in lanes:
            s = ln["st"].cuda_stream
            ln["ext"].extract_batch_dev(ln["fr"][k & 1], ln["kps"][cur], ln["desc"][cur], ln["n"][cur], stream=s)
            api.ORBmatcher.best2_batch_dev(ln["desc"][cur], ln["n"][cur], ln["desc"][prv], ln["n"][prv], ln["idx"], ln["best"], ln["second"],
                                           ln["cap"], device=0, stream=s)
    for k in range(3):
        step(k)
    torch.cuda.synchronize()
    if stagger and L > 1:   # lane l starts l / L of a step late
        for l, ln in enumerate(lanes):
            if l:
                with torch.cuda.stream(ln["st"]):
                    torch.cuda._sleep(int(stagger * l / L * 2.0e3))   # cycles
    t0 = time.perf_counter()
    for k in range(3, 3 + STEPS):
        step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(int(ln["n"][(2 + STEPS) % 3].min()) >= NFEAT for ln in lanes)
    print("lanes %d (x %d frames) stagger %4d us: %.1f us per step of %d frames, %.0f frames/s  ok=%s" % (L, b, stagger, 1e6 * dt / STEPS, B, B * STEPS / dt, ok), flush=True)
    for ln in lanes:
        del ln["ext"]


for L, stg in ((1, 0), (2, 0), (2, 150), (4, 0), (4, 300), (8, 0)):
    if B % L == 0:
        run(L, stg)
