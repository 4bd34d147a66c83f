"""Diagnostic (CPU, numpy + the oracle): how selective FAST pre-tests are on the synthetic frames, per threshold:
   exact FAST-9 corners, the 4-point compass test (c4), compass + diagonal compass, the 8-point test (c8), popcount >= 9;
   pixel level and dword (4 pixels) level.   python tools/diag/fast_stats.py 640 480 1000"""
import sys, numpy as np
sys.path.insert(0, '.')
from oracle import bindings as ob
from weiner_slamit_v2_amd import synth
W, H = int(sys.argv[1]), int(sys.argv[2]); NF = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
ring = [(0,3),(1,3),(2,2),(3,1),(3,0),(3,-1),(2,-2),(1,-3),(0,-3),(-1,-3),(-2,-2),(-3,-1),(-3,0),(-3,1),(-2,2),(-1,3)]
def flags(img, t):
    h, w = img.shape
    c = img[3:h-3, 3:w-3].astype(np.int16)
    B = []; D = []
    for dx, dy in ring:
        r = img[3+dy:h-3+dy, 3+dx:w-3+dx].astype(np.int16)
        B.append(r > c + t); D.append(r < c - t)
    return np.array(B), np.array(D)
def contig(F, n, step=1):
    K = F.shape[0]
    out = np.zeros(F.shape[1:], bool)
    for k in range(K):
        a = np.ones(F.shape[1:], bool)
        for j in range(n): a &= F[(k + j) % K]
        out |= a
    return out
o = ob.OrbOracle(NF)
tot = {}
for idx in range(2):
    img = synth.synth_frame(W, H, idx)
    o.extract(img)
    for lv in range(8):
        L = o.level(lv)
        for t in (20, 7):
            B, D = flags(L, t)
            exact = contig(B, 9) | contig(D, 9)
            c4 = contig(B[::4], 2) | contig(D[::4], 2)
            c8 = contig(B[::2], 4) | contig(D[::2], 4)
            # 12 of 16? any-12-point-subsets: skip. "compass4 AND odd-compass4"
            c4b = contig(B[2::4], 2) | contig(D[2::4], 2)
            # count of bright >= 9 or dark >= 9 (popcount test)
            pc = (B.sum(0) >= 9) | (D.sum(0) >= 9)
            for name, m in (('exact', exact), ('c4', c4), ('c4&c4diag', c4 & c4b), ('c8', c8), ('pop9', pc), ('c8&pop9', c8 & pc)):
                k = (t, name)
                a = tot.setdefault(k, [0, 0]); a[0] += int(m.sum()); a[1] += m.size
            # dword-level (groups of 4 px along x) survival
            for name, m in (('exact', exact), ('c4', c4), ('c8', c8)):
                hh, ww = m.shape; w4 = ww // 4 * 4
                g = m[:, :w4].reshape(hh, w4 // 4, 4).any(2)
                a = tot.setdefault((t, name + '_dw'), [0, 0]); a[0] += int(g.sum()); a[1] += g.size
for k in sorted(tot): print(k, '%.4f' % (tot[k][0] / tot[k][1]))
