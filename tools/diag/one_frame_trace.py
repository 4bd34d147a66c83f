"""Kernel timeline of single-frame extract calls (run under rocprofv3 --kernel-trace): python3 tools/diag/one_frame_trace.py"""
import sys
sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, synth
img = synth.synth_frame(640, 480, 1)
ext = api.ORBextractor(1000, 1.2, 8, 20, 7)
for _ in range(12):
    ext(img)
