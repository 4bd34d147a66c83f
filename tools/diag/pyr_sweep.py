"""Stage timings of the extractor for the current SLAMIT_PYR_SEGS / SLAMIT_PYR_TILE (diagnostic knobs)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, synth  # noqa: E402

B = 64
frames = synth.synth_batch(640, 480, B)
ext = api.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
ext.extract_batch(frames)
ext.profile(True)
for _ in range(10):
    ext.extract_batch(frames)
p = ext.profile(True)
print({k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in p.items()})
