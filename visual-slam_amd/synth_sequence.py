"""Synthetic EuRoC-layout stereo sequences for the headless pipeline (there is no network, so no EuRoC
download): a textured box room rendered by ray casting through the double-sphere model of the reference's
calibration files (calibration_file/euroc_v1_123_ds_calib.json layout), a smooth camera path, ground
truth in the `state_groundtruth_estimate0/data.csv` format the reference reads
(include/io/dataset_io_euroc.h:83-110), `cam{0,1}/data.csv` with the CRLF lines src/slam.cpp:1006-1040
expects, and 8-bit grey PNGs written with zlib (which the C++ side decodes with its own inflate).

Test infrastructure and benchmark input only -- the product path never imports this module.
"""
import json
import os
import struct
import zlib

import numpy as np

W, H = 752, 480

# the reference's EuRoC calibration values (double sphere); cam0 = body frame
CALIB = {
    "T_i_c": [
        dict(px=0.0, py=0.0, pz=0.0, qx=0.0, qy=0.0, qz=0.0, qw=1.0),
        dict(px=0.11002674958788125, py=-0.0002891377986657201, pz=0.00024662504991979133,
             qx=0.007123658988066061, qy=0.0006289220699998059, qz=0.0010774952115908369, qw=0.9999738481299002),
    ],
    "intrinsics": [
        dict(cam_type="ds", fx=351.037283216868, fy=350.00745559773659, cx=365.8880973548215, cy=249.34573836993605,
             p1=-0.23853128172699646, p2=0.5678694845290938, p3=0.0, p4=0.0, width=W, height=H),
        dict(cam_type="ds", fx=362.9532887030661, fy=361.85685537441409, cx=379.35501913798876, cy=256.0392416777184,
             p1=-0.21063783723054772, p2=0.5776109411992846, p3=0.0, p4=0.0, width=W, height=H),
    ],
}


def write_calibration(path):
    """cereal JSON layout of include/visnav/serialization.h:113-167."""
    doc = {"value0": {"cam.T_i_c": CALIB["T_i_c"], "cam.intrinsics": CALIB["intrinsics"]}}
    with open(path, "w") as f:
        json.dump(doc, f, indent=4)


# ------------------------------------------------------------------------------------------ PNG
def _chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def write_png(path, img, level=6, filters="none", idat_split=0):
    """8-bit grey (H, W) or RGB (H, W, 3) PNG.  filters: 'none' | 'mixed' (cycles through all five PNG
    filter types, exercising the decoder's unfilter code)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ch = 1 if img.ndim == 2 else img.shape[2]
    ctype = {1: 0, 3: 2, 4: 6}[ch]
    rows = img.reshape(h, w * ch).astype(np.int16)
    raw = bytearray()
    prev = np.zeros(w * ch, np.int16)
    for y in range(h):
        cur = rows[y]
        ft = 0 if filters == "none" else y % 5
        a = np.concatenate([np.zeros(ch, np.int16), cur[:-ch]])
        b = prev
        c = np.concatenate([np.zeros(ch, np.int16), prev[:-ch]])
        if ft == 0:
            out = cur
        elif ft == 1:
            out = cur - a
        elif ft == 2:
            out = cur - b
        elif ft == 3:
            out = cur - ((a + b) >> 1)
        else:
            p = a + b - c
            pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
            out = cur - pred
        raw.append(ft)
        raw += (out & 0xFF).astype(np.uint8).tobytes()
        prev = cur
    comp = zlib.compress(bytes(raw), level)
    parts = [comp] if not idat_split else [comp[i:i + idat_split] for i in range(0, len(comp), idat_split)]
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(_chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)))
        for p in parts:
            f.write(_chunk(b"IDAT", p))
        f.write(_chunk(b"IEND", b""))


def write_pgm(path, img):
    img = np.ascontiguousarray(img, np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


# ------------------------------------------------------------------------------------------ geometry
def quat_to_rot(q):  # x y z w
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def rot_to_quat(R):  # -> x y z w
    w = np.sqrt(max(0.0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
    if w > 1e-6:
        return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])
    x = np.sqrt(max(0.0, 1 + R[0, 0] - R[1, 1] - R[2, 2])) / 2
    return np.array([x, (R[0, 1] + R[1, 0]) / (4 * x), (R[0, 2] + R[2, 0]) / (4 * x), (R[2, 1] - R[1, 2]) / (4 * x)])


def rot_y(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def rot_x(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def ds_unproject_grid(intr):
    """Unit bearing of every pixel centre (H, W, 3): camera_models.h:272-302."""
    fx, fy, cx, cy, xi, alpha = (intr[k] for k in ("fx", "fy", "cx", "cy", "p1", "p2"))
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    mx, my = (u - cx) / fx, (v - cy) / fy
    rr = mx * mx + my * my
    mz = (1 - alpha * alpha * rr) / (alpha * np.sqrt(1 - (2 * alpha - 1) * rr) + 1 - alpha)
    s = (mz * xi + np.sqrt(mz * mz + (1 - xi * xi) * rr)) / (mz * mz + rr)
    return np.stack([mx * s, my * s, mz * s - xi], axis=-1)


def make_texture(rng, size=1024):
    """Corner-rich texture: smooth background + many random rectangles, lightly blurred."""
    low = rng.random((size // 64 + 1, size // 64 + 1))
    ys = np.linspace(0, low.shape[0] - 1.001, size)
    xs = np.linspace(0, low.shape[1] - 1.001, size)
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    bg = (low[y0][:, x0] * (1 - fy) * (1 - fx) + low[y0 + 1][:, x0] * fy * (1 - fx) + low[y0][:, x0 + 1] * (1 - fy) * fx +
          low[y0 + 1][:, x0 + 1] * fy * fx)
    tex = 70 + 90 * bg
    n_rect = 2200
    cx = rng.integers(0, size, n_rect)
    cy = rng.integers(0, size, n_rect)
    w = rng.integers(6, 48, n_rect)
    h = rng.integers(6, 48, n_rect)
    val = rng.integers(15, 240, n_rect)
    for i in range(n_rect):
        tex[max(0, cy[i] - h[i] // 2):cy[i] + h[i] // 2, max(0, cx[i] - w[i] // 2):cx[i] + w[i] // 2] = val[i]
    # 3x3 box blur (anti-aliasing for the bilinear lookups)
    p = np.pad(tex, 1, mode="edge")
    tex = sum(p[dy:dy + size, dx:dx + size] for dy in range(3) for dx in range(3)) / 9.0
    return tex.astype(np.float32)


class BoxRoom:
    """Axis-aligned room, the camera moves inside; every wall carries its own texture."""

    def __init__(self, seed, half=(4.0, 2.5, 4.0), px_per_m=110.0):
        rng = np.random.default_rng(seed)
        self.half = np.asarray(half, np.float64)
        self.px_per_m = px_per_m
        self.tex = [make_texture(rng) for _ in range(6)]

    def render(self, bearings, R_wc, c):
        d = bearings.reshape(-1, 3) @ R_wc.T  # world ray directions
        best_t = np.full(d.shape[0], np.inf)
        best_face = np.zeros(d.shape[0], np.int32)
        for axis in range(3):
            for sgn, face in ((-1.0, 2 * axis), (1.0, 2 * axis + 1)):
                with np.errstate(divide="ignore", invalid="ignore"):
                    t = (sgn * self.half[axis] - c[axis]) / d[:, axis]
                ok = (t > 1e-6) & (t < best_t)
                best_t = np.where(ok, t, best_t)
                best_face = np.where(ok, face, best_face)
        hit = c[None, :] + best_t[:, None] * d
        out = np.zeros(d.shape[0], np.float32)
        for face in range(6):
            m = best_face == face
            if not m.any():
                continue
            axis = face // 2
            a0, a1 = [a for a in range(3) if a != axis]
            tex = self.tex[face]
            size = tex.shape[0]
            u = (hit[m, a0] + self.half[a0]) * self.px_per_m
            v = (hit[m, a1] + self.half[a1]) * self.px_per_m
            u = np.clip(u, 0, size - 1.001)
            v = np.clip(v, 0, size - 1.001)
            u0, v0 = u.astype(np.int32), v.astype(np.int32)
            fu, fv = (u - u0).astype(np.float32), (v - v0).astype(np.float32)
            out[m] = (tex[v0, u0] * (1 - fu) * (1 - fv) + tex[v0, u0 + 1] * fu * (1 - fv) + tex[v0 + 1, u0] * (1 - fu) * fv +
                      tex[v0 + 1, u0 + 1] * fu * fv)
        return np.clip(np.rint(out), 0, 255).astype(np.uint8).reshape(bearings.shape[:2])


def trajectory(n_frames, step_m=0.03, radius=1.2, look_deg=0.0):
    """Body (= cam0) poses T_w_i: a circle in the x-z plane (y is down), looking along the tangent with a
    slow pitch oscillation; ~3 cm and ~1.4 degrees per frame.  look_deg turns the viewing direction about the
    vertical axis relative to the tangent (+90: radially outwards, towards the nearest wall -- close,
    well-conditioned structure for an 11 cm stereo baseline; 0: along the path, structure 4-7 m away)."""
    poses = []
    for k in range(n_frames):
        th = k * step_m / radius
        c = np.array([radius * np.cos(th), 0.15 * np.sin(2.3 * th), radius * np.sin(th)])
        # forward = tangent direction (-sin, 0, cos): yaw such that z_cam maps to it
        yaw = np.arctan2(-np.sin(th), np.cos(th)) + np.deg2rad(look_deg)
        R = rot_y(yaw) @ rot_x(0.06 * np.sin(1.7 * th))
        poses.append((R, c))
    return poses


_RENDER = {}


def _render_frame(k):
    room, bear, T_i_c, poses, out_dir, stamps, png_level = (_RENDER[x] for x in ("room", "bear", "T_i_c", "poses", "out_dir", "stamps", "png_level"))
    R_wi, t_wi = poses[k]
    for c in range(2):
        R_wc = R_wi @ T_i_c[c][0]
        t_wc = R_wi @ T_i_c[c][1] + t_wi
        img = room.render(bear[c], R_wc, t_wc)
        write_png(os.path.join(out_dir, "cam%d" % c, "data", "%d.png" % stamps[k]), img, level=png_level)
    return k


def render_sequence(out_dir, n_frames=60, seed=1, t0_ns=1403715273262142976, dt_ns=50_000_000, png_level=1,
                    step_m=0.03, radius=1.2, workers=1, look_deg=0.0, room_half=(4.0, 2.5, 4.0), px_per_m=110.0):
    """Writes <out_dir>/{cam0,cam1}/data.csv + data/*.png, state_groundtruth_estimate0/data.csv and
    <out_dir>/calib.json.  Returns the list of body poses (R_wi, t_wi).  The path is a circle: more than
    2 pi radius / step_m frames revisit the start (loop closure).  workers > 1 renders frames in forked
    processes (call before anything touches the GPU)."""
    room = BoxRoom(seed, half=room_half, px_per_m=px_per_m)
    poses = trajectory(n_frames, step_m=step_m, radius=radius, look_deg=look_deg)
    bear = [ds_unproject_grid(CALIB["intrinsics"][c]) for c in range(2)]
    T_i_c = []
    for c in range(2):
        t = CALIB["T_i_c"][c]
        T_i_c.append((quat_to_rot(np.array([t["qx"], t["qy"], t["qz"], t["qw"]])), np.array([t["px"], t["py"], t["pz"]])))
    for c in range(2):
        os.makedirs(os.path.join(out_dir, "cam%d" % c, "data"), exist_ok=True)
    os.makedirs(os.path.join(out_dir, "state_groundtruth_estimate0"), exist_ok=True)
    stamps = [t0_ns + k * dt_ns for k in range(n_frames)]
    for c in range(2):
        with open(os.path.join(out_dir, "cam%d" % c, "data.csv"), "w", newline="") as f:
            f.write("#timestamp [ns],filename\r\n")
            for s in stamps:
                f.write("%d,%d.png\r\n" % (s, s))
    _RENDER.update(room=room, bear=bear, T_i_c=T_i_c, poses=poses, out_dir=out_dir, stamps=stamps, png_level=png_level)
    if workers > 1:
        import multiprocessing as mp
        with mp.get_context("fork").Pool(workers) as pool:
            pool.map(_render_frame, range(n_frames), chunksize=4)
    else:
        for k in range(n_frames):
            _render_frame(k)
    # ground truth at 4x the frame rate (positions interpolated on the same parametrisation), EuRoC columns
    fine = trajectory(4 * (n_frames - 1) + 1, step_m=step_m / 4, radius=radius, look_deg=look_deg)
    with open(os.path.join(out_dir, "state_groundtruth_estimate0", "data.csv"), "w", newline="") as f:
        f.write("#timestamp, p_RS_R_x [m], p_RS_R_y [m], p_RS_R_z [m], q_RS_w [], q_RS_x [], q_RS_y [], q_RS_z [], "
                "v_RS_R_x, v_RS_R_y, v_RS_R_z, b_w_x, b_w_y, b_w_z, b_a_x, b_a_y, b_a_z\r\n")
        for j, (R, t) in enumerate(fine):
            q = rot_to_quat(R)
            f.write("%d,%.9f,%.9f,%.9f,%.9f,%.9f,%.9f,%.9f,0,0,0,0,0,0,0,0,0\r\n" %
                    (t0_ns + j * (dt_ns // 4), t[0], t[1], t[2], q[3], q[0], q[1], q[2]))
    write_calibration(os.path.join(out_dir, "calib.json"))
    return poses
