#!/usr/bin/env python3
"""Writes tests/golden/orb_{vga,vga2000,720p}.npz: the CPU restatement's (oracle/orb_oracle.cc) outputs on three
synthetic frames, frozen as REGRESSION GUARDS.

NON-AUTHORITATIVE: the ORB half is "parity unpinned" (OpenCV 2.4.9 is not available here and the reference holds no
fixtures for this path), so these vectors pin the restatement against accidental edits, not against the reference.
The frames come from weiner_slamit_v2_amd.synth (deterministic); only their CRC-32 is stored.

    python tools/gen_orb_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bindings as ob  # noqa: E402
from weiner_slamit_v2_amd import synth  # noqa: E402

CASES = {"vga": (640, 480, 1000, 3), "vga2000": (640, 480, 2000, 4), "720p": (1280, 720, 2000, 6)}


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def golden(width, height, nfeatures, index):
    img = synth.synth_frame(width, height, index)
    orc = ob.OrbOracle(nfeatures, 1.2, 8, 20, 7)
    kps, desc = orc.extract(img)
    out = {"params": np.array([width, height, nfeatures, index, 8, 20, 7], np.int32), "img_crc": np.uint32(crc(img)),
           "desc": desc, "angle_bits": kps["angle"].view(np.uint32).copy()}
    for f in ("x", "y", "size", "response", "octave"):
        out["kp_" + f] = np.ascontiguousarray(kps[f])
    cand_n, cand_crc, lvl_crc, blur_crc, lvl_kps = [], [], [], [], []
    for l in range(8):
        c = orc.candidates(l)
        cand_n.append(len(c)); cand_crc.append(crc(c)); lvl_crc.append(crc(orc.level(l)))
        b = orc.blurred(l)
        blur_crc.append(crc(b) if b is not None else 0)
        lvl_kps.append(int((kps["octave"] == l).sum()))
    out.update(cand_n=np.array(cand_n, np.int32), cand_crc=np.array(cand_crc, np.uint32), level_crc=np.array(lvl_crc, np.uint32),
               blur_crc=np.array(blur_crc, np.uint32), level_kps=np.array(lvl_kps, np.int32))
    return out


if __name__ == "__main__":
    for name, (w, h, nf, idx) in CASES.items():
        g = golden(w, h, nf, idx)
        path = os.path.join(ROOT, "tests", "golden", "orb_%s.npz" % name)
        np.savez_compressed(path, **g)
        print(path, len(g["desc"]), "keypoints", os.path.getsize(path), "bytes")
