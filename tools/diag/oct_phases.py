"""Per-phase ticks of octree_kernel per (frame, level) workgroup (build with -DOCT_DIAG)."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from weiner_slamit_v2_amd import api, synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = synth.synth_batch(640, 480, B)
ext = api.ORBextractor(1000, 1.2, 8, 20, 7, max_batch=B)
ext.extract_batch(frames)
ext.extract_batch(frames)
out = np.zeros(4096 * 8, np.uint64)
api.lib().slamit_diag_oct(out.ctypes.data_as(C.c_void_p))
r = out.reshape(-1, 8)[:B * 8].reshape(8, B, 8).transpose(1, 0, 2)   # the grid is (frames, levels): workgroup w = level * frames + frame
t0 = r[:, :, 7].min()
print("realtime (100 MHz): first start 0, last start %.1f us, last end %.1f us" % ((r[:, :, 7].max() - t0) / 100.0, (r[:, :, 5].max() - t0) / 100.0))
for lvl in (0, 7):
    print("level %d WG duration %.1f us (mean), start offset mean %.1f us" % (lvl, ((r[:, lvl, 5] - r[:, lvl, 7]) / 100.0).mean(), ((r[:, lvl, 7] - t0) / 100.0).mean()))
names = ["roots", "childcnt sweep", "careful rank", "flags+scan+nodes", "relabel sweep", "best+out"]
for lvl in range(8):
    x = r[:, lvl].astype(np.float64)
    print("level %d: keys %5.0f passes %4.1f nodes %4.0f  total %7.0f ticks | " % (
        lvl, np.mean(r[:, lvl, 6] & np.uint64(0xFFFFFFFF)), np.mean(r[:, lvl, 6] >> np.uint64(32)), 0.0, x[:, :5].sum(1).mean()) +
        "  ".join("%s %.0f" % (n, x[:, i].mean()) for i, n in enumerate(names)))
