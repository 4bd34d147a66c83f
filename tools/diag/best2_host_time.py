"""Latency of slamit_hamming_best2 / slamit_frame_finish through host pointers."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, synth  # noqa: E402
rs = np.random.RandomState(1)
a, b = rs.randint(0, 256, (1000, 32)).astype(np.uint8), rs.randint(0, 256, (1000, 32)).astype(np.uint8)
api.ORBmatcher.best2(a, b)
r = []
for _ in range(200):
    t0 = time.perf_counter(); api.ORBmatcher.best2(a, b); r.append(time.perf_counter() - t0)
r.sort(); print("best2 1000x1000: median %.3f ms min %.3f ms" % (1e3 * r[100], 1e3 * r[0]))
ext = api.ORBextractor(1000, 1.2, 8, 20, 7)
k, d = ext(synth.synth_frame(640, 480, 1))
cam = [526.69, 540.36, 313.07, 238.39, 0.262383, -0.953104, -0.005358, 0.002628, 1.163314]
args = (cam, k, -4.3, -2.7, 64 / 649.4, 48 / 486.6)
api.Frame.finish(*args)
r = []
for _ in range(200):
    t0 = time.perf_counter(); api.Frame.finish(*args); r.append(time.perf_counter() - t0)
r.sort(); print("frame_finish 1000 kps: median %.3f ms min %.3f ms" % (1e3 * r[100], 1e3 * r[0]))
