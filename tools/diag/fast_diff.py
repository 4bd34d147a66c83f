"""Diagnostic: FAST candidate lists of the device vs the oracle, level by level; prints the differing entries.
   python tools/diag/fast_diff.py [W H nfeatures nframes]"""
import sys
import numpy as np
sys.path.insert(0, '.')
from oracle import bindings as ob
from weiner_slamit_v2_amd import api, synth

W, H, NF, N = (int(a) for a in (sys.argv[1:5] + ['640', '480', '1000', '3'][len(sys.argv) - 1:]))
ext, orc = api.ORBextractor(NF, 1.2, 8, 20, 7), ob.OrbOracle(NF)
bad = 0
for idx in range(N):
    img = synth.synth_frame(W, H, idx) if idx < N - 1 else synth.noise_frame(W, H)
    ext(img); orc.extract(img)
    for l in range(8):
        cg, co = ext.debug_candidates(0, l), orc.candidates(l)
        sg = set(map(tuple, cg.tolist())); so = set(map(tuple, co.tolist()))
        if sg != so or len(cg) != len(co):
            bad += 1
            print("frame %d level %d: device %d oracle %d | only device %s | only oracle %s" % (
                idx, l, len(cg), len(co), sorted(sg - so)[:12], sorted(so - sg)[:12]))
        elif not np.array_equal(cg, co):
            bad += 1
            print("frame %d level %d: same set, different order" % (idx, l))
print("mismatching (frame, level) pairs:", bad)
