"""Per-phase wave cycles of fast_cells_kernel (diagnostic build with -DFAST_DIAG):
    tools/diag/build_diag.sh fast orb_kernels.hip -DFAST_DIAG
    SLAMIT_LIB=$PWD/tools/diag/libdiag_fast.so python tools/diag/fast_phases.py
"""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, synth  # noqa: E402

B = 64
frames = synth.synth_batch(640, 480, B)
ext = api.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
ext.extract_batch(frames)
ext.profile(True)
L = api.lib()
out = np.zeros(65536 * 8, np.uint64)
L.slamit_diag_fast(out.ctypes.data_as(C.c_void_p), 1)
ext.extract_batch(frames)
L.slamit_diag_fast(out.ctypes.data_as(C.c_void_p), 0)
print("stage us:", ext.profile(True))
r = out.reshape(-1, 8)
r = r[r[:, 7] > 0]
ph = r[:, :5].astype(np.float64)
names = ["stage", "pretest+compact", "score", "nms", "output"]
tot = ph.sum()
for i, n in enumerate(names):
    print("%-16s %14.0f ticks  %5.1f %%   median/wave %8.0f" % (n, ph[:, i].sum(), 100 * ph[:, i].sum() / tot, np.median(ph[:, i])))
surv = (r[:, 5] >> np.uint64(32)).sum()
npx = (r[:, 5] & np.uint64(0xFFFFFFFF)).sum()
span = float(r[:, 7].max() - r[:, 6].min())
print("waves %d  scan px %d  survivors %d (%.2f %%)  ticks/wave %.0f  kernel span %.0f ticks  mean resident waves %.0f"
      % (len(r), npx, surv, 100.0 * surv / npx, tot / len(r), span, tot / span))
