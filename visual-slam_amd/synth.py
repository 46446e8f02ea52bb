"""Seeded synthetic inputs shaped like the reference's data (no dataset is reachable offline).

* stereo_pair(): 752x480 8-bit stereo images of a layered random-rectangle world -- thousands of
  corners of comparable strength, so that goodFeaturesToTrack's 1 % quality level still leaves
  >= 1500 corners (EuRoC frames give 350-700), plus per-layer disparity so that left/right
  descriptors really match.  BASELINE.json configs[1] ("Synthetic 752x480 stereo, 1500 feats").
* ba_problem(): a local / global bundle-adjustment problem with the EuRoC double-sphere
  intrinsics (calibration_file/euroc_v1_123_ds_calib.json of the reference), BASELINE.json
  configs[2] and [4].
* vocabulary_text(): a random k-ary vocabulary tree in the ORBvoc.txt text format
  (TemplatedVocabulary.h:1338-1424; the real ORBvoc.txt is a missing blob).
"""
import numpy as np

W, H = 752, 480

# calibration_file/euroc_v1_123_ds_calib.json (fx fy cx cy xi alpha 0 0) and T_i_c of cam1
DS_INTR = np.array([
    [351.037283216868, 350.00745559773659, 365.8880973548215, 249.34573836993605,
     -0.23853128172699646, 0.5678694845290938, 0.0, 0.0],
    [362.9532887030661, 361.85685537441409, 379.35501913798876, 256.0392416777184,
     -0.21063783723054772, 0.5776109411992846, 0.0, 0.0]])
T_0_1 = np.array([0.007123658988066061, 0.0006289220699998059, 0.0010774952115908369,
                  0.9999738481299002, 0.11002674958788125, -0.0002891377986657201,
                  0.00024662504991979133])  # qx qy qz qw tx ty tz


def _blur3(img):
    k = np.array([0.25, 0.5, 0.25])
    p = np.pad(img, 1, mode="reflect")
    t = p[:, :-2] * k[0] + p[:, 1:-1] * k[1] + p[:, 2:] * k[2]
    return t[:-2] * k[0] + t[1:-1] * k[1] + t[2:] * k[2]


def stereo_scene(seed, w=W, h=H, n_rects=2600, margin=0):
    """The noise-free float images (left, right) of one random-rectangle world.  `margin` > 0 keeps a band of
    that many pixels along the image border free of rectangles (smooth background only)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    bg = 110 + 30 * np.sin(xx / 97.0 + rng.uniform(0, 6)) * np.cos(yy / 71.0 + rng.uniform(0, 6))
    left = bg.astype(np.float64).copy()
    right = bg.astype(np.float64).copy()
    # painter's order: far first
    depth = np.sort(rng.uniform(2.0, 12.0, n_rects))[::-1]
    fb = DS_INTR[0, 0] * 0.55 * T_0_1[4]  # effective focal (ds model, centre) x baseline
    lo_x, hi_x, lo_y, hi_y = margin, w - margin, margin, h - margin
    for z in depth:
        rw, rh = rng.integers(7, 30, 2)
        x0 = int(rng.integers(-10, w))
        y0 = int(rng.integers(-10, h))
        g = float(rng.choice([rng.uniform(20, 90), rng.uniform(150, 235)]))
        d = int(round(fb / z))
        ya, yb = max(lo_y, y0), min(hi_y, y0 + rh)
        if ya >= yb:
            continue
        xa, xb = max(lo_x, x0), min(hi_x, x0 + rw)
        if xa < xb:
            left[ya:yb, xa:xb] = g
        xa, xb = max(lo_x, x0 - d), min(hi_x, x0 + rw - d)
        if xa < xb:
            right[ya:yb, xa:xb] = g
    return _blur3(left), _blur3(right), rng


def _to_u8(im, rng, noise):
    return np.clip(np.rint(im + rng.normal(0, noise, im.shape)), 0, 255).astype(np.uint8)


def stereo_pair(seed, w=W, h=H, n_rects=2600, noise=2.0, margin=0):
    """Returns (left, right) uint8 images.

    margin = 0: corners everywhere, so goodFeaturesToTrack's 1500 strongest include ~12 % inside the 19-px border
    that detectKeypoints then drops (keypoints.h:145-149): ~1320 keypoints per image.  margin = 24 (bench.py):
    the border band holds no rectangle edges, the 1500 strongest corners are interior and ~1500 survive -- the
    "1500 feats/frame" of BASELINE configs[1]."""
    left, right, rng = stereo_scene(seed, w, h, n_rects, margin)
    return _to_u8(left, rng, noise), _to_u8(right, rng, noise)


def stereo_pair_variants(seed, n_variants, w=W, h=H, n_rects=2600, noise=2.0, margin=0):
    """n_variants distinct stereo pairs of ONE scene: independent sensor-noise realisations (every pixel differs,
    so do the corner responses, their order and the selected set).  Variant 0 == stereo_pair(seed, ...).
    Returns uint8 array (n_variants, 2, h, w)."""
    left, right, rng = stereo_scene(seed, w, h, n_rects, margin)
    out = np.empty((n_variants, 2, h, w), np.uint8)
    for v in range(n_variants):
        if v:
            rng = np.random.default_rng([seed, v])
        out[v, 0] = _to_u8(left, rng, noise)
        out[v, 1] = _to_u8(right, rng, noise)
    return out


def random_descriptors(rng, n):
    return rng.integers(0, 2 ** 63, size=(n, 4), dtype=np.int64).astype(np.uint64) * np.uint64(2) + \
        rng.integers(0, 2, size=(n, 4), dtype=np.int64).astype(np.uint64)


def flip_bits(rng, desc, nbits):
    """Copy of `desc` (n,4 u64) with `nbits` random bit flips per row."""
    out = desc.copy()
    for i in range(len(out)):
        for b in rng.choice(256, size=nbits, replace=False):
            out[i, b // 64] ^= np.uint64(1) << np.uint64(b % 64)
    return out


# ------------------------------------------------------------------------------------- geometry
def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def quat_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def se3_mul(a, b):
    q = quat_mul(a[:4], b[:4])
    t = a[4:] + quat_R(a[:4]) @ b[4:]
    return np.concatenate([q / np.linalg.norm(q), t])


def axis_angle_q(axis, ang):
    axis = np.asarray(axis, float)
    axis = axis / np.linalg.norm(axis)
    return np.concatenate([axis * np.sin(ang / 2), [np.cos(ang / 2)]])


def ds_project(intr, p):
    fx, fy, cx, cy, xi, alpha = intr[:6]
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    d1 = np.sqrt(x * x + y * y + z * z)
    k = xi * d1 + z
    d2 = np.sqrt(x * x + y * y + k * k)
    den = alpha * d2 + (1 - alpha) * k
    return np.stack([fx * x / den + cx, fy * y / den + cy], -1), den


def ba_problem(seed, n_kf=7, n_lms=20000, pix_noise=0.5, outlier_frac=0.05, pose_noise=(0.02, 0.0087),
               point_noise=0.05, n_fixed_kf=1, max_range=15.0, w=W, h=H, loop_radius=None,
               integer_pixels=True):
    """Stereo keyframes on a smooth trajectory looking at a random point cloud.

    Returns a dict of numpy arrays in the flattened layout of include/vslam_hip.h (cameras 2*k and
    2*k+1 are the left/right camera of keyframe k; the first n_fixed_kf keyframes are fixed, like
    the reference fixes the oldest keyframe's two cameras, src/slam.cpp:1545-1551).
    """
    rng = np.random.default_rng(seed)
    n_cams = 2 * n_kf
    poses = np.zeros((n_cams, 7))
    if loop_radius is None:
        # local window: a short arc, 0.05 m and 0.5 deg per frame, a keyframe every ~8 frames
        for k in range(n_kf):
            s = 8 * k
            q = axis_angle_q([0, 1, 0], np.deg2rad(0.5 * s))
            t = np.array([0.05 * s, 0.02 * np.sin(0.3 * s), 0.01 * s])
            poses[2 * k] = np.concatenate([q, t])
            poses[2 * k + 1] = se3_mul(poses[2 * k], T_0_1)
        centre = poses[n_cams // 2, 4:] + quat_R(poses[n_cams // 2, :4]) @ np.array([0, 0, 6.0])
        pts = centre + rng.uniform(-1, 1, (n_lms, 3)) * np.array([8.0, 4.0, 5.0])
    else:
        # global: keyframes on a circle looking outward at a cylindrical shell of points
        for k in range(n_kf):
            a = 2 * np.pi * k / n_kf
            q = axis_angle_q([0, 1, 0], a)
            t = np.array([loop_radius * np.sin(a), 0.05 * np.sin(5 * a), loop_radius * np.cos(a)])
            poses[2 * k] = np.concatenate([q, t])
            poses[2 * k + 1] = se3_mul(poses[2 * k], T_0_1)
        a = rng.uniform(0, 2 * np.pi, n_lms)
        r = loop_radius + rng.uniform(3.0, 9.0, n_lms)
        pts = np.stack([r * np.sin(a), rng.uniform(-2.5, 2.5, n_lms), r * np.cos(a)], -1)
    cam_intr = np.tile(np.array([0, 1], np.int32), n_kf)
    obs_cam, obs_lm, obs_uv = [], [], []
    for c in range(n_cams):
        R = quat_R(poses[c, :4])
        pc = (pts - poses[c, 4:]) @ R  # R^T (p - t)
        uv, den = ds_project(DS_INTR[cam_intr[c]], pc)
        rngm = np.linalg.norm(pc, axis=1)
        ok = (pc[:, 2] > 0.3) & (den > 0) & (uv[:, 0] >= 19) & (uv[:, 0] < w - 19) & \
             (uv[:, 1] >= 19) & (uv[:, 1] < h - 19) & (rngm < max_range)
        idx = np.nonzero(ok)[0]
        obs_cam.append(np.full(len(idx), c, np.int32))
        obs_lm.append(idx.astype(np.int32))
        obs_uv.append(uv[idx])
    obs_cam = np.concatenate(obs_cam)
    obs_lm = np.concatenate(obs_lm)
    obs_uv = np.concatenate(obs_uv)
    # keep landmarks with >= 2 observations, renumber
    cnt = np.bincount(obs_lm, minlength=n_lms)
    keep = cnt >= 2
    remap = -np.ones(n_lms, np.int64)
    remap[keep] = np.arange(keep.sum())
    sel = keep[obs_lm]
    obs_cam, obs_lm, obs_uv = obs_cam[sel], remap[obs_lm[sel]].astype(np.int32), obs_uv[sel]
    pts = pts[keep]
    # order observations by landmark then camera (the order the reference adds residual blocks:
    # for each landmark, for each obs, map_utils.h:369-395)
    order = np.lexsort((obs_cam, obs_lm))
    obs_cam, obs_lm, obs_uv = obs_cam[order], obs_lm[order], obs_uv[order]
    # detected corners are integer pixels in the reference (goodFeaturesToTrack has no sub-pixel step)
    noise = rng.normal(0, pix_noise, obs_uv.shape)
    out = rng.random(len(obs_uv)) < outlier_frac
    noise[out] = rng.normal(0, 20.0, (out.sum(), 2))
    obs_uv = obs_uv + noise
    if integer_pixels:
        obs_uv = np.rint(obs_uv)
    gt_poses, gt_pts = poses.copy(), pts.copy()
    cam_fixed = np.zeros(n_cams, np.uint8)
    cam_fixed[:2 * n_fixed_kf] = 1
    for c in range(n_cams):
        if cam_fixed[c]:
            continue
        dq = axis_angle_q(rng.normal(size=3), rng.normal(0, pose_noise[1]))
        poses[c] = np.concatenate([quat_mul(poses[c, :4], dq), poses[c, 4:] + rng.normal(0, pose_noise[0], 3)])
        poses[c, :4] /= np.linalg.norm(poses[c, :4])
    pts = pts + rng.normal(0, point_noise, pts.shape)
    return dict(poses=poses, cam_fixed=cam_fixed, cam_intr=cam_intr, intr=DS_INTR.copy(), points=pts,
                obs_cam=obs_cam, obs_lm=obs_lm, obs_uv=obs_uv, cam_model=(0, 0), gt_poses=gt_poses,
                gt_points=gt_pts)


# ----------------------------------------------------------------------------------- vocabulary
def vocabulary_text(seed, k=10, L=3, stop_frac=0.02):
    """Text of a random k-ary, depth-L vocabulary in the ORBvoc.txt format:
    header `k L scoring weighting` (0 0 = L1_NORM, TF_IDF), then one line per node in creation
    order (breadth-first): `parent is_leaf d0 .. d31 weight`.  Children descriptors are their
    parent's descriptor with a shrinking number of random bit flips, so descents are meaningful.
    A few leaves get weight 0 ("stopped" words, TemplatedVocabulary.h:1158)."""
    rng = np.random.default_rng(seed)
    lines = ["%d %d 0 0" % (k, L)]
    level = [(0, rng.integers(0, 256, 32, dtype=np.uint8))]
    next_id = 1
    for lev in range(1, L + 1):
        nxt = []
        flips = max(8, 96 >> (lev - 1))
        for pid, pdesc in level:
            for _ in range(k):
                d = pdesc.copy()
                for b in rng.choice(256, size=flips, replace=False):
                    d[b // 8] ^= np.uint8(1 << (b % 8))
                leaf = 1 if lev == L else 0
                wgt = 0.0
                if leaf:
                    wgt = 0.0 if rng.random() < stop_frac else float(rng.uniform(0.5, 9.0))
                lines.append("%d %d %s %.6f" % (pid, leaf, " ".join(str(int(v)) for v in d), wgt))
                nxt.append((next_id, d))
                next_id += 1
        level = nxt
    return "\n".join(lines) + "\n"


def vocabulary_arrays(seed, k=10, L=6, stop_frac=0.02):
    """The same kind of tree as vocabulary_text() as arrays, vectorised so that the ORB-SLAM shape (k = 10, L = 6:
    1,111,111 nodes, 1,000,000 words -- what ORBvoc.txt holds, TemplatedVocabulary.h:1338-1424; the file itself is a
    missing blob) is generated in seconds.  Breadth-first creation order: the k children of a node are consecutive.
    Returns parent [n] i32, is_leaf [n] u8, desc [n, 32] u8, weight [n] f64 for nodes 1..n (node 0 = root is implicit)."""
    rng = np.random.default_rng(seed)
    parents, leaves, descs, weights = [], [], [], []
    level_ids = np.zeros(1, np.int64)
    level_desc = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    next_id = 1
    for lev in range(1, L + 1):
        flips = max(8, 96 >> (lev - 1))
        m = len(level_ids) * k
        bits = np.unpackbits(np.repeat(level_desc, k, axis=0), axis=1, bitorder="little")       # [m, 256]
        # `flips` random positions per child (a repeated position flips once: a child differs in <= flips bits)
        pos = rng.integers(0, 256, (m, flips))
        np.put_along_axis(bits, pos, 1 - np.take_along_axis(bits, pos, axis=1), axis=1)
        d = np.packbits(bits, axis=1, bitorder="little")
        leaf = lev == L
        w = np.zeros(m)
        if leaf:
            w = np.where(rng.random(m) < stop_frac, 0.0, rng.uniform(0.5, 9.0, m))
            w = np.round(w, 6)    # what the text format's %.6f keeps
        parents.append(np.repeat(level_ids, k).astype(np.int32))
        leaves.append(np.full(m, 1 if leaf else 0, np.uint8))
        descs.append(d)
        weights.append(w)
        level_ids = np.arange(next_id, next_id + m, dtype=np.int64)
        level_desc = d
        next_id += m
    return (np.concatenate(parents), np.concatenate(leaves), np.concatenate(descs, axis=0), np.concatenate(weights))


def write_vocabulary_text(path, k, L, parent, is_leaf, desc, weight, chunk=65536):
    """ORBvoc.txt layout (`k L 0 0`, then `parent is_leaf d0 .. d31 weight` per node), written in chunks."""
    byte_txt = np.array([str(i) for i in range(256)])
    with open(path, "w") as f:
        f.write("%d %d 0 0\n" % (k, L))
        for a in range(0, len(parent), chunk):
            b = min(a + chunk, len(parent))
            dtxt = byte_txt[desc[a:b]]
            rows = [" ".join((str(p), str(lf), " ".join(dr), "%.6f" % w))
                    for p, lf, dr, w in zip(parent[a:b].tolist(), is_leaf[a:b].tolist(), dtxt.tolist(), weight[a:b].tolist())]
            f.write("\n".join(rows))
            f.write("\n")


def pose_graph(seed, n_nodes=60, n_loop_edges=25, meas_noise=0.0, drift=0.02, outlier_edges=0, window=0):
    """Keyframe poses on a noisy loop + relative-pose edges (loop_closure_utils.h:446-587: spanning-tree edges
    between consecutive keyframes, covisibility edges, one loop constraint).  Returns a dict with
    poses_gt / poses (drifted initial guess) [N, 7] (qx qy qz qw tx ty tz), node_fixed [N], edge_a, edge_b [E]
    and edge_meas [E, 6] = log(T_a^-1 T_b) of the ground truth (+ noise), the functor's upsilon_omega."""
    rng = np.random.default_rng(seed)
    gt = []
    for k in range(n_nodes):
        th = 2 * np.pi * k / n_nodes
        c = np.array([3.0 * np.cos(th), 0.3 * np.sin(3 * th), 3.0 * np.sin(th)])
        q = axis_angle_q(np.array([0.1 * np.sin(th), 1.0, 0.1 * np.cos(2 * th)]), th + 0.5 * np.pi)
        gt.append(np.concatenate([q, c]))
    gt = np.array(gt)

    def inv(p):
        R = quat_R(p[:4])
        qi = np.array([-p[0], -p[1], -p[2], p[3]])
        return np.concatenate([qi, -R.T @ p[4:]])

    def mul(a, b):
        ax, ay, az, aw = a[:4]
        bx, by, bz, bw = b[:4]
        q = np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                      aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])
        return np.concatenate([q / np.linalg.norm(q), quat_R(a[:4]) @ b[4:] + a[4:]])

    def log(p):  # numpy restatement of Sophus::SE3::log for the fixture (independent of the oracle's C++)
        q, t = p[:4], p[4:]
        n = np.linalg.norm(q[:3])
        if n < 1e-12:
            om = 2.0 / q[3] * q[:3]
        else:
            half = np.arctan2(n, q[3]) if q[3] >= 0 else np.arctan2(-n, -q[3])
            om = 2.0 * half / n * q[:3]
        th = np.linalg.norm(om)
        Om = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
        if th < 1e-8:
            Vi = np.eye(3) - 0.5 * Om + Om @ Om / 12.0
        else:
            Vi = np.eye(3) - 0.5 * Om + (1 - th * np.cos(th / 2) / (2 * np.sin(th / 2))) / th ** 2 * (Om @ Om)
        return np.concatenate([Vi @ t, om])

    ea, eb = [], []
    for k in range(n_nodes - 1):
        ea.append(k + 1)   # the functor's T_w_c is the newer keyframe, T_w_n the older one
        eb.append(k)
    for _ in range(n_loop_edges):
        a = int(rng.integers(2, n_nodes))
        b = int(rng.integers(0, a - 1))
        ea.append(a)
        eb.append(b)
    # covisibility edges the way the reference's pose graph has them (keyframes within `window` of each other in time
    # order, and around the closed loop): a graph whose normal equations are a narrow CYCLIC band
    for j in range(2, window + 1):
        for k in range(n_nodes):
            a, b = k, k - j
            if b < 0:
                a, b = k - j + n_nodes, k      # across the seam of the loop (newer keyframe first)
                if a <= b + 1:
                    continue
            ea.append(a)
            eb.append(b)
    ea.append(n_nodes - 1)  # the loop constraint
    eb.append(0)
    meas = np.array([log(mul(inv(gt[a]), gt[b])) for a, b in zip(ea, eb)])
    meas = meas + meas_noise * rng.normal(size=meas.shape)
    for k in rng.choice(len(ea), outlier_edges, replace=False) if outlier_edges else []:
        meas[k, :3] += rng.normal(0, 2.0, 3)   # gross outliers for the Huber loss
    # drifted initial guess: integrate perturbed relative poses
    poses = [gt[0].copy()]
    for k in range(1, n_nodes):
        rel = mul(inv(gt[k - 1]), gt[k])
        rel[4:] += drift * rng.normal(size=3)
        dq = axis_angle_q(rng.normal(size=3), drift * 0.3 * rng.normal())
        rel = mul(rel, np.concatenate([dq, np.zeros(3)]))
        poses.append(mul(poses[-1], rel))
    fixed = np.zeros(n_nodes, np.uint8)
    fixed[n_nodes - 1] = 1  # LoopClosureOptions::set_current_kf_fixed
    return dict(poses_gt=gt, poses=np.array(poses), node_fixed=fixed, edge_a=np.array(ea, np.int32), edge_b=np.array(eb, np.int32),
                edge_meas=meas, log=log, mul=mul, inv=inv)
