"""PCIe-inclusive rates of the host-pointer entry points (upload + kernels + download)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, synth  # noqa: E402

img = synth.synth_frame(640, 480, 1)
ext = api.ORBextractor(1000, 1.2, 8, 20, 7)
ext(img)
t0 = time.perf_counter()
for _ in range(50):
    k, d = ext(img)
t1 = (time.perf_counter() - t0) / 50
frames = synth.synth_batch(640, 480, 64)
ext64 = api.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=64)
ext64.extract_batch(frames)
t0 = time.perf_counter()
for _ in range(10):
    ks, ds = ext64.extract_batch(frames)
t64 = (time.perf_counter() - t0) / 10
k2, d2 = ext(synth.warp_frame(img, 1))
t0 = time.perf_counter()
for _ in range(50):
    api.ORBmatcher.best2(d, d2)
tm = (time.perf_counter() - t0) / 50
print("extract 1 VGA frame, host pointers: %.3f ms (%.0f frames/s); batch of 64: %.2f ms (%.0f frames/s); best2 1000x1000 host pointers: %.3f ms" % (
    1e3 * t1, 1 / t1, 1e3 * t64, 64 / t64, 1e3 * tm))
