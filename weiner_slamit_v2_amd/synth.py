"""Deterministic synthetic inputs for the hot path.

Frames: 3-octave value noise + high-contrast axis-aligned / rotated rectangles and checker
patches + uniform noise, so every pyramid level holds far more FAST-20 corners than its quota
and some cells need the FAST-7 fallback.  Everything is drawn from numpy's frozen legacy
MT19937 stream (RandomState) and built from exact float64 elementwise arithmetic, so the same
seed gives the same bytes on any host.

BA windows: the 50-keyframe x 2000-point problem of SURVEY.md §8d (arc trajectory, reference
intrinsics of Tracking.cc:77-80, float32-rounded inputs widened to float64 the way
Converter.cc:37-47,110-116 does).
"""
import numpy as np

FRAME_SEED = 0xC0FFEE


def _value_noise(rs, h, w, cell, amp):
    gh, gw = h // cell + 2, w // cell + 2
    g = rs.uniform(-1.0, 1.0, size=(gh, gw))
    ys = np.arange(h, dtype=np.float64) / cell
    xs = np.arange(w, dtype=np.float64) / cell
    y0 = np.floor(ys).astype(np.int64)
    x0 = np.floor(xs).astype(np.int64)
    fy = (ys - y0)[:, None]
    fx = (xs - x0)[None, :]
    a = g[y0][:, x0]
    b = g[y0][:, x0 + 1]
    c = g[y0 + 1][:, x0]
    d = g[y0 + 1][:, x0 + 1]
    return amp * ((a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy)


def synth_frame(width, height, index=0, n_shapes=None, seed=FRAME_SEED, scene="rich"):
    """uint8 (height, width) frame number `index`.  scene "rich" (every test, the benchmarks' default): value noise down to
    4-pixel cells plus 400 hard-edged shapes per VGA frame -- about one pixel in eight is a FAST corner at threshold 20;
    "sparse": the same construction without the 4-pixel noise level and with 150 shapes -- a few percent of corners, closer
    to indoor video (bench.py --scene sparse reports how the FAST pass depends on it)."""
    rs = np.random.RandomState((seed + index) & 0x7FFFFFFF)
    img = np.full((height, width), 128.0)
    sparse = scene == "sparse"
    for cell, amp in (((64, 64.0), (16, 16.0)) if sparse else ((64, 64.0), (16, 32.0), (4, 16.0))):
        img += _value_noise(rs, height, width, cell, amp)
    if n_shapes is None:
        n_shapes = int(round((150.0 if sparse else 400.0) * (width * height) / (640.0 * 480.0)))
    yy, xx = np.mgrid[0:height, 0:width]
    for _ in range(n_shapes):
        kind = rs.randint(0, 3)
        cx, cy = rs.randint(0, width), rs.randint(0, height)
        sw, sh = rs.randint(6, 41), rs.randint(6, 41)
        delta = float(rs.randint(60, 121)) * (1.0 if rs.randint(0, 2) else -1.0)
        x0, x1 = max(cx - 30, 0), min(cx + 31, width)
        y0, y1 = max(cy - 30, 0), min(cy + 31, height)
        lx = (xx[y0:y1, x0:x1] - cx).astype(np.float64)
        ly = (yy[y0:y1, x0:x1] - cy).astype(np.float64)
        if kind == 0:  # axis-aligned rectangle
            m = (np.abs(lx) * 2 <= sw) & (np.abs(ly) * 2 <= sh)
            img[y0:y1, x0:x1] += delta * m
        elif kind == 1:  # rotated rectangle
            th = rs.uniform(0.0, np.pi)
            c, s = np.cos(th), np.sin(th)
            u, v = c * lx + s * ly, -s * lx + c * ly
            m = (np.abs(u) * 2 <= sw) & (np.abs(v) * 2 <= sh)
            img[y0:y1, x0:x1] += delta * m
        else:  # 2x2 checker patch
            m = (np.abs(lx) * 2 <= sw) & (np.abs(ly) * 2 <= sh)
            sign = np.where((lx >= 0) ^ (ly >= 0), 1.0, -1.0)
            img[y0:y1, x0:x1] += delta * m * sign
    img += rs.randint(-2, 3, size=(height, width)) if sparse else rs.randint(-4, 5, size=(height, width))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def warp_frame(frame, index=0, seed=FRAME_SEED):
    """Frame B = frame A under a known similarity (rot +-10 deg, scale 0.9-1.1, shift <= 20 px),
    nearest-neighbour sampled with integer-exact index maps, + noise +-2."""
    h, w = frame.shape
    rs = np.random.RandomState((seed ^ 0x5EED) + index)
    ang = rs.uniform(-10.0, 10.0) * np.pi / 180.0
    sc = rs.uniform(0.9, 1.1)
    tx, ty = rs.uniform(-20, 20, size=2)
    yy, xx = np.mgrid[0:h, 0:w]
    cx, cy = (w - 1) / 2.0, (h - 1) / 2.0
    c, s = np.cos(ang) / sc, np.sin(ang) / sc
    sx = c * (xx - cx - tx) + s * (yy - cy - ty) + cx
    sy = -s * (xx - cx - tx) + c * (yy - cy - ty) + cy
    ix = np.clip(np.rint(sx).astype(np.int64), 0, w - 1)
    iy = np.clip(np.rint(sy).astype(np.int64), 0, h - 1)
    out = frame[iy, ix].astype(np.int64) + rs.randint(-2, 3, size=(h, w))
    return np.clip(out, 0, 255).astype(np.uint8)


def synth_batch(width, height, n, first=0, seed=FRAME_SEED):
    return np.stack([synth_frame(width, height, first + i, seed=seed) for i in range(n)])


def flat_frame(width, height, value=90):
    return np.full((height, width), value, dtype=np.uint8)


def noise_frame(width, height, seed=1):
    return np.random.RandomState(seed).randint(0, 256, size=(height, width)).astype(np.uint8)


# ---------------------------------------------------------------------------------------------
# Bundle-adjustment windows
# ---------------------------------------------------------------------------------------------

INTRINSICS = (526.69, 540.36, 313.07, 238.39)  # Tracking.cc:77-80


def _hat(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def se3_exp(xi):
    """xi = [omega(3), upsilon(3)] -> (R, t), same closed form as g2o SE3Quat::exp."""
    w, u = np.asarray(xi[:3], float), np.asarray(xi[3:], float)
    th = np.linalg.norm(w)
    W = _hat(w)
    if th < 1e-5:
        R = np.eye(3) + W + W @ W
        V = R
    else:
        W2 = W @ W
        R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * W2
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * W + (th - np.sin(th)) / th ** 3 * W2
    return R, V @ u


def _inv_sigma2_table(nlevels=8, scale=1.2):
    s = np.float32(1.0)
    out = []
    for _ in range(nlevels):
        out.append(np.float32(1.0) / (s * s))
        s = np.float32(s * np.float64(np.float32(scale)))
    return np.array(out, dtype=np.float32)


def synth_ba(n_kf=50, n_pt=2000, obs_per_pt=8, outlier_frac=0.03, seed=12345, n_fixed=1, stereo_frac=0.0, baseline=0.08):
    """Returns a dict of numpy arrays laid out as include/slamit.h:slamit_ba_problem wants them.

    obs_per_pt = None or >= n_kf gives the dense pattern (every point in every keyframe).
    stereo_frac > 0: that fraction of the observations also carries the keypoint's column in the right image (edge_ur; -1 on the
    monocular ones, KeyFrame::mvuRight) and the dict gains kf_bf = baseline x fx per keyframe (KeyFrame::mbf): the window of a
    stereo / RGB-D session, Optimizer.cc:621-650.  The monocular arrays do not depend on stereo_frac (own random stream)."""
    rs = np.random.RandomState(seed)
    fx, fy, cx, cy = INTRINSICS
    Rs, ts = [], []
    for k in range(n_kf):
        R, t = se3_exp([0.0, 0.02 * k, 0.0, -0.1 * k, 0.0, 0.0])
        Rs.append(R)
        ts.append(t)
    pts = np.stack([rs.uniform(0.5, 4.5, n_pt), rs.uniform(-1.5, 1.5, n_pt), rs.uniform(4.0, 8.0, n_pt)], 1)
    dense = obs_per_pt is None or obs_per_pt >= n_kf
    inv_sig = _inv_sigma2_table()
    quota = np.array([217, 181, 151, 126, 105, 87, 73, 60], dtype=np.float64)
    quota /= quota.sum()
    e_kf, e_pt, e_uv, e_is, e_ur = [], [], [], [], []
    rs2 = np.random.RandomState(seed + 777)
    bf = float(np.float32(baseline * fx))
    for p in range(n_pt):
        if dense:
            kfs = range(n_kf)
        else:
            start = rs.randint(0, n_kf - obs_per_pt + 1)
            kfs = range(start, start + obs_per_pt)
        for k in kfs:
            Xc = Rs[k] @ pts[p] + ts[k]
            u = fx * Xc[0] / Xc[2] + cx + rs.normal(0.0, 1.0)
            v = fy * Xc[1] / Xc[2] + cy + rs.normal(0.0, 1.0)
            if rs.uniform() < outlier_frac:
                u += rs.choice([-30.0, 30.0])
                v += rs.choice([-30.0, 30.0])
            octave = rs.choice(8, p=quota)
            e_kf.append(k)
            e_pt.append(p)
            e_uv.append((u, v))
            e_is.append(inv_sig[octave])
            if stereo_frac > 0:
                ur = u - bf / Xc[2] + rs2.normal(0.0, 1.0)
                if rs2.uniform() < outlier_frac:
                    ur += rs2.choice([-20.0, 20.0])
                e_ur.append(ur if rs2.uniform() < stereo_frac else -1.0)
    # perturbed initial estimates, rounded to float32 then widened (Converter.cc)
    poses = np.zeros((n_kf, 12))
    for k in range(n_kf):
        if k < n_fixed:
            R, t = Rs[k], ts[k]
        else:
            dR, dt = se3_exp(rs.normal(0.0, 0.005, 6))
            R, t = dR @ Rs[k], dR @ ts[k] + dt
        poses[k, :9] = R.reshape(-1)
        poses[k, 9:] = t
    poses = poses.astype(np.float32).astype(np.float64)
    pts0 = (pts + rs.normal(0.0, 0.02, pts.shape)).astype(np.float32).astype(np.float64)
    fixed = np.zeros(n_kf, dtype=np.uint8)
    fixed[:n_fixed] = 1
    intr = np.tile(np.array([fx, fy, cx, cy], dtype=np.float32).astype(np.float64), (n_kf, 1))
    extra = {}
    if stereo_frac > 0:
        extra = {"edge_ur": np.array(e_ur, dtype=np.float32).astype(np.float64), "kf_bf": np.full(n_kf, bf)}
    return {
        **extra,
        "kf_pose": np.ascontiguousarray(poses),
        "kf_fixed": fixed,
        "kf_intr": np.ascontiguousarray(intr),
        "pt_xyz": np.ascontiguousarray(pts0),
        "edge_kf": np.array(e_kf, dtype=np.int32),
        "edge_pt": np.array(e_pt, dtype=np.int32),
        "edge_uv": np.array(e_uv, dtype=np.float32).astype(np.float64),
        "edge_inv_sigma2": np.array(e_is, dtype=np.float32).astype(np.float64),
        "truth_pose": np.array([np.concatenate([Rs[k].reshape(-1), ts[k]]) for k in range(n_kf)]),
        "truth_pt": pts,
    }


def synth_pose(n=400, outlier_frac=0.15, seed=7, perturb=0.02, stereo_frac=0.0, baseline=0.08):
    """One PoseOptimization problem (Optimizer.cc:239-451): n map points seen by one frame, pixel
    noise N(0,1), a fraction of gross outliers, initial pose = truth perturbed by exp(N(0, perturb^2)).
    All inputs float32-rounded then widened, as the reference feeds them."""
    rs = np.random.RandomState(seed)
    fx, fy, cx, cy = INTRINSICS
    R, t = se3_exp([0.05, -0.1, 0.02, 0.3, -0.1, 0.2])
    pts_c = np.stack([rs.uniform(-2.5, 2.5, n), rs.uniform(-1.8, 1.8, n), rs.uniform(3.0, 9.0, n)], 1)
    xw = (pts_c - t) @ R  # X_w = R^T (X_c - t)
    u = fx * pts_c[:, 0] / pts_c[:, 2] + cx + rs.normal(0, 1, n)
    v = fy * pts_c[:, 1] / pts_c[:, 2] + cy + rs.normal(0, 1, n)
    bad = rs.uniform(size=n) < outlier_frac
    u[bad] += rs.choice([-1, 1], bad.sum()) * rs.uniform(8, 60, bad.sum())
    v[bad] += rs.choice([-1, 1], bad.sum()) * rs.uniform(8, 60, bad.sum())
    quota = np.array([217, 181, 151, 126, 105, 87, 73, 60], dtype=np.float64)
    inv_sig = _inv_sigma2_table()[rs.choice(8, n, p=quota / quota.sum())]
    dR, dt = se3_exp(rs.normal(0, perturb, 6))
    pose = np.concatenate([(dR @ R).reshape(-1), dR @ t + dt])
    f32 = lambda a: np.ascontiguousarray(np.asarray(a, np.float32).astype(np.float64))
    extra = {}
    if stereo_frac > 0:   # a stereo / RGB-D frame: that fraction of the keypoints has a right-image column (own random stream)
        rs2 = np.random.RandomState(seed + 777)
        bf = float(np.float32(baseline * fx))
        ur = u - bf / pts_c[:, 2] + rs2.normal(0, 1, n)
        bad_r = rs2.uniform(size=n) < outlier_frac
        ur[bad_r] += rs2.choice([-1, 1], bad_r.sum()) * rs2.uniform(8, 40, bad_r.sum())
        ur = np.where(rs2.uniform(size=n) < stereo_frac, ur, -1.0)
        extra = {"ur": f32(ur), "bf": bf}
    return {**extra, "pose": f32(pose), "intr": f32([fx, fy, cx, cy]), "xw": f32(xw), "uv": f32(np.stack([u, v], 1)),
            "inv_sigma2": f32(inv_sig), "truth_pose": np.concatenate([R.reshape(-1), t]), "truth_outlier": bad}


def synth_search(n_kp=1500, m=600, seed=0, width=640, height=480, th=3.0, crowd=False, retarget=True):
    """A guided-search problem in the shape Tracking::SearchLocalPoints hands to
    ORBmatcher::SearchByProjection: a frame's undistorted keypoints + grid constants, and map-point
    queries projected near some of them (several queries may aim at the same keypoint, which is what
    makes the reference's loop order-dependent).  Returns (frame, queries) dicts for api / oracle."""
    rs = np.random.RandomState(1000 + seed)
    # undistorted image bounds are slightly outside the sensor (Frame::ComputeImageBounds)
    min_x, max_x, min_y, max_y = np.float32(-4.3), np.float32(width + 5.1), np.float32(-2.7), np.float32(height + 3.9)
    inv_w = np.float32(64) / np.float32(max_x - min_x)   # Frame.cc:90-91
    inv_h = np.float32(48) / np.float32(max_y - min_y)
    span = 60.0 if crowd else None
    if crowd:
        xy = np.stack([rs.uniform(300, 300 + span, n_kp), rs.uniform(200, 200 + span, n_kp)], 1).astype(np.float32)
    else:
        xy = np.stack([rs.uniform(min_x - 2, max_x + 2, n_kp), rs.uniform(min_y - 2, max_y + 2, n_kp)], 1).astype(np.float32)
    octave = rs.randint(0, 8, n_kp).astype(np.int32)
    desc = rs.randint(0, 256, (n_kp, 32)).astype(np.uint8)
    taken = (rs.rand(n_kp) < 0.15).astype(np.uint8)
    scale = np.float32(1.2) ** np.arange(8, dtype=np.float32)
    tgt = rs.randint(0, max(n_kp, 1), m) if n_kp else np.zeros(m, np.int64)
    if retarget:
        tgt[m // 2:] = tgt[:m - m // 2][rs.permutation(m - m // 2)]    # second half re-targets the first half (worst case for the ordered walk)
    uvr = np.zeros((m, 3), np.float32)
    lmin, lmax = np.zeros(m, np.int32), np.zeros(m, np.int32)
    qdesc = np.zeros((m, 32), np.uint8)
    for q in range(m):
        k = int(tgt[q]) if n_kp else 0
        lvl = int(octave[k]) if n_kp else 0
        pl = int(np.clip(lvl + rs.randint(-1, 2), 0, 7))                 # predicted level
        r = np.float32(th) * np.float32(rs.choice([2.5, 4.0])) * scale[pl]
        base = xy[k] if n_kp else np.zeros(2, np.float32)
        uvr[q] = (base[0] + rs.uniform(-0.7, 0.7) * r, base[1] + rs.uniform(-0.7, 0.7) * r, r)
        mode = rs.randint(0, 4)
        if mode == 0:
            lmin[q], lmax[q] = pl - 1, pl                                # SearchByProjection(F, MPs): [level-1, level]
        elif mode == 1:
            lmin[q], lmax[q] = pl, -1                                    # frame-to-frame, forward motion
        elif mode == 2:
            lmin[q], lmax[q] = 0, pl                                     # backward motion
        else:
            lmin[q], lmax[q] = pl - 1, pl + 1
        d = (desc[k] if n_kp else np.zeros(32, np.uint8)).copy()
        flips = rs.randint(0, 70)                                        # up to ~70 flipped bits around the target
        for b in rs.randint(0, 256, flips):
            d[b >> 3] ^= np.uint8(1 << (b & 7))
        qdesc[q] = d
    valid = (rs.rand(m) < 0.9).astype(np.uint8)
    takes = (rs.rand(m) < 0.95).astype(np.uint8)
    # a few windows entirely outside the grid
    for q in range(0, m, 37):
        uvr[q, 0] = max_x + 500.0 if (q // 37) % 2 else min_x - 500.0
    frame = dict(kp_xy=xy, kp_octave=octave, desc=desc, kp_taken=taken, min_x=float(min_x), min_y=float(min_y),
                 inv_w=float(inv_w), inv_h=float(inv_h))
    queries = dict(uvr=uvr, level_min=lmin, level_max=lmax, desc=qdesc, valid=valid, takes=takes)
    return frame, queries


def synth_init_pair(n=1500, seed=0):
    """Two frames for ORBmatcher::SearchForInitialization: F2's keypoints are F1's moved by a few pixels (plus clutter),
    descriptors a few bits apart, so that several F1 keypoints compete for one F2 keypoint (the take-over path)."""
    rs = np.random.RandomState(4000 + seed)
    f2, _ = synth_search(n, 4, 50 + seed)
    f2 = dict(f2)
    f2["kp_octave"] = np.where(rs.rand(n) < 0.7, 0, rs.randint(1, 8, n)).astype(np.int32)
    n1 = n
    src = rs.randint(0, n, n1)
    src[n1 // 2:] = src[:n1 - n1 // 2]                      # pairs of F1 keypoints aiming at the same F2 keypoint
    o1 = np.where(rs.rand(n1) < 0.75, 0, rs.randint(1, 8, n1)).astype(np.int32)
    prev = (f2["kp_xy"][src] + rs.uniform(-25, 25, (n1, 2))).astype(np.float32)
    d1 = f2["desc"][src].copy()
    for j in range(n1):
        for b in rs.randint(0, 256, rs.randint(0, 45)):
            d1[j, b >> 3] ^= np.uint8(1 << (b & 7))
    f1 = dict(kp_octave=o1, desc=d1, angle=rs.uniform(0, 360, n1).astype(np.float32))
    f2["angle"] = rs.uniform(0, 360, n).astype(np.float32)
    return f1, prev, f2


def synth_bow(n1=1000, n2=1000, n_nodes=100, seed=0, mode=0, big_group=0):
    """Two keyframes as the BoW drivers see them: descriptors, per-feature vocabulary node (DBoW2::FeatureVector at
    levelsup 4 puts ~10 features of a 1000-feature frame into each of ~100 nodes), validity masks, and the node groups
    common to both sides in ascending node order (what the reference's while / lower_bound walk visits).  Side 2 holds
    noisy copies of side-1 descriptors (several per source now and then, so queries compete for candidates) that mostly
    share the source's node.  mode 1 adds the SearchForTriangulation geometry: a pure sideways translation between two
    identical pinhole cameras (epipolar lines are image rows), matching features on the same row up to a pixel or two.
    big_group > 0 forces that many candidates into one node (more than one wavefront of them).
    Returns (side1, side2, groups, epi)."""
    rs = np.random.RandomState(5000 + seed)
    d1 = rs.randint(0, 256, (n1, 32)).astype(np.uint8)
    node1 = rs.randint(0, max(n_nodes, 1), n1)
    src = rs.randint(0, max(n1, 1), n2) if n1 else np.zeros(n2, np.int64)
    d2 = d1[src].copy() if n1 else rs.randint(0, 256, (n2, 32)).astype(np.uint8)
    for i in range(n2):
        for b in rs.randint(0, 256, rs.randint(0, 90)):          # up to ~90 flipped bits
            d2[i, b >> 3] ^= np.uint8(1 << (b & 7))
    node2 = np.where(rs.rand(n2) < 0.85, node1[src] if n1 else 0, rs.randint(0, max(n_nodes, 1), n2))
    if big_group and n2:
        node2[:min(big_group, n2)] = node1[0] if n1 else 0
    if n2 > 10:                                                   # exact duplicates: ties in distance
        d2[1] = d2[0]; node2[1] = node2[0]
    side1 = dict(desc=d1, valid=(rs.rand(n1) < 0.8).astype(np.uint8))
    side2 = dict(desc=d2, valid=None if mode == 0 and seed % 2 == 0 else (rs.rand(n2) < 0.85).astype(np.uint8))
    qp, qi, cp, ci = [0], [], [0], []
    for node in sorted(set(node1.tolist()) & set(node2.tolist())):
        a, b = np.nonzero(node1 == node)[0], np.nonzero(node2 == node)[0]
        qi += a.tolist(); ci += b.tolist()
        qp.append(len(qi)); cp.append(len(ci))
    groups = dict(q_ptr=np.array(qp, np.int32), q_idx=np.array(qi, np.int32), c_ptr=np.array(cp, np.int32), c_idx=np.array(ci, np.int32))
    epi = None
    if mode == 1:
        fx, fy, cx, cy = np.float32(517.3), np.float32(516.5), np.float32(318.6), np.float32(255.3)
        xy1 = np.stack([rs.uniform(20, 620, n1), rs.uniform(20, 460, n1)], 1).astype(np.float32)
        # camera 2 = camera 1 shifted along x: x2 = x1 - disparity, y2 = y1 (+ noise; some far off the line)
        disp = rs.uniform(2, 60, n2).astype(np.float32)
        noise = np.where(rs.rand(n2) < 0.75, rs.normal(0, 0.8, n2), rs.normal(0, 12.0, n2)).astype(np.float32)
        xy2 = np.stack([xy1[src, 0] - disp, xy1[src, 1] + noise], 1).astype(np.float32) if n1 else np.zeros((n2, 2), np.float32)
        oct2 = rs.randint(0, 8, n2).astype(np.int32)
        # F12 with x1^T F12 x2 = 0 for t = (b, 0, 0), R = I:  F = K^-T [t]x K^-1  (any scale)
        K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float64)
        tx = np.array([[0, 0, 0], [0, 0, -1.0], [0, 1.0, 0]])
        Ki = np.linalg.inv(K)
        F = (Ki.T @ tx @ Ki).astype(np.float32)
        if seed % 3 == 0:
            F[:, :] = 0                                           # degenerate: den == 0 for every query
        scale = (np.float32(1.2) ** np.arange(8, dtype=np.float32)).astype(np.float32)
        # the epipole of a sideways translation is at infinity; put a finite one inside the image to exercise :737-743
        epi = dict(F12=F.reshape(9), ex=float(xy2[0, 0]) if n2 else 0.0, ey=float(xy2[0, 1]) if n2 else 0.0,
                   scale_factor=scale.tolist(), level_sigma2=(scale * scale).tolist())
        side1["kp_xy"] = xy1
        side2["kp_xy"] = xy2
        side2["kp_octave"] = oct2
    return side1, side2, groups, epi


def synth_sim3(n=200, outlier_frac=0.15, seed=0, perturb=0.03, fix_scale=False, scale=1.08):
    """An Optimizer::OptimizeSim3 problem as the reference assembles it (Optimizer.cc:1099-1178): the same physical points as
    map points of two keyframes, each in its own camera frame (p1, p2, float values), their keypoints in both images, a true
    similarity p1 = s R p2 + t and a perturbed initial g2oS12 (what Sim3Solver's RANSAC hands over).  Outliers are wrong
    keypoint associations.  Returns a dict in the layout of slamit_sim3_problem."""
    rs = np.random.RandomState(9000 + seed)
    f32 = np.float32
    intr1 = np.array([517.3, 516.5, 318.6, 255.3], f32).astype(np.float64)
    intr2 = np.array([520.9, 521.0, 325.1, 249.7], f32).astype(np.float64)
    s_true = 1.0 if fix_scale else scale
    R, t = se3_exp(np.array([0.05, -0.08, 0.03, 0.4, -0.1, 0.2]))
    p2 = np.stack([rs.uniform(-2.5, 2.5, n), rs.uniform(-1.8, 1.8, n), rs.uniform(2.5, 9.0, n)], 1)
    p1 = s_true * (p2 @ R.T) + t + rs.normal(0, 0.01, (n, 3))           # two independent estimates of the same points
    p1[:, 2] = np.maximum(p1[:, 2], 0.5)
    p1, p2 = p1.astype(f32).astype(np.float64), p2.astype(f32).astype(np.float64)
    obs1 = np.stack([intr1[0] * p1[:, 0] / p1[:, 2] + intr1[2], intr1[1] * p1[:, 1] / p1[:, 2] + intr1[3]], 1) + rs.normal(0, 0.7, (n, 2))
    obs2 = np.stack([intr2[0] * p2[:, 0] / p2[:, 2] + intr2[2], intr2[1] * p2[:, 1] / p2[:, 2] + intr2[3]], 1) + rs.normal(0, 0.7, (n, 2))
    bad = rs.rand(n) < outlier_frac
    obs1[bad] += rs.uniform(-60, 60, (int(bad.sum()), 2))
    scale_f = f32(1.2) ** np.arange(8, dtype=f32)
    lv1, lv2 = rs.randint(0, 8, n), rs.randint(0, 8, n)
    isig1 = (f32(1) / (scale_f * scale_f))[lv1].astype(np.float64)
    isig2 = (f32(1) / (scale_f * scale_f))[lv2].astype(np.float64)
    dR, dt = se3_exp(rs.normal(0, perturb, 6))
    R0, t0 = dR @ R, dR @ t + dt
    s0 = s_true * (1.0 if fix_scale else float(np.exp(rs.normal(0, perturb))))
    return dict(n=n, p1=p1, p2=p2, obs1=obs1.astype(f32).astype(np.float64), obs2=obs2.astype(f32).astype(np.float64),
                inv_sigma2_1=isig1, inv_sigma2_2=isig2, intr1=intr1, intr2=intr2, r12=R0.reshape(9), t12=t0, s12=s0,
                th2=10.0, fix_scale=int(fix_scale), true=dict(R=R, t=t, s=s_true, bad=bad))
