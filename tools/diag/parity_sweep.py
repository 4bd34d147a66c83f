"""One-off wide parity sweep of the extractor against the oracle (sizes, feature counts, level counts, scale factors)."""
import itertools
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import bindings as ob  # noqa: E402
from weiner_slamit_v2_amd import api, synth  # noqa: E402

bad = 0
cases = [(640, 480, 1000, 8, 1.2), (752, 480, 1500, 8, 1.2), (1241, 376, 2000, 8, 1.2), (1920, 1080, 3000, 8, 1.2),
         (640, 480, 500, 4, 1.5), (640, 480, 5000, 8, 1.2), (1280, 720, 2000, 12, 1.15), (800, 600, 1200, 5, 2.0),
         (333, 777, 700, 6, 1.3), (1024, 1024, 2500, 7, 1.25), (640, 480, 1000, 1, 1.2), (2048, 1536, 4000, 8, 1.2)]
for (w, h, nf, nl, sf) in cases:
    try:
        ext, orc = api.ORBextractor(nf, sf, nl, 20, 7), ob.OrbOracle(nf, sf, nl, 20, 7)
    except Exception as e:
        print("create failed", (w, h, nf, nl, sf), e)
        continue
    for seed, kind in ((1, "synth"), (2, "noise")):
        img = synth.synth_frame(w, h, seed) if kind == "synth" else synth.noise_frame(w, h, seed)
        try:
            kg, dg = ext(img)
        except Exception as e:
            print("extract failed", (w, h, nf, nl, sf, kind), e)
            bad += 1
            continue
        ko, do = orc.extract(img)
        same = len(kg) == len(ko) and all(np.array_equal(kg[f], ko[f]) for f in ("x", "y", "octave", "response", "size")) and \
            np.array_equal(kg["angle"].view(np.uint32), ko["angle"].view(np.uint32)) and np.array_equal(dg, do)
        print("%-34s %-5s n=%5d %s" % ((w, h, nf, nl, sf), kind, len(kg), "ok" if same else "MISMATCH"))
        bad += not same
print("mismatches:", bad)
sys.exit(1 if bad else 0)
