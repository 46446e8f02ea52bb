"""CPU checks of the headless pipeline's host code (include/visnav_amd/harness/): the C++ test program
tests/cpp/harness_host_test.cpp is compiled with g++ and run on files written here -- PNGs produced by
zlib at several compression levels / filter mixes (the C++ side has its own inflate), the EuRoC csv
layout with CRLF lines (src/slam.cpp:1006-1040), the cereal calibration JSON, and a trajectory whose
alignment error is computed independently with numpy (src/slam.cpp:1618-1710)."""
import importlib
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def _numpy_ate(te, pe, tg, pg):
    est, gt = [], []
    for t, p in zip(te, pe):
        j = int(np.searchsorted(tg, t, side="right")) - 1
        if j < 0 or j >= len(tg) - 1:
            continue
        dt = float(tg[j + 1] - tg[j])
        if dt > 1.1e8:
            continue
        r = float(t - tg[j]) / dt
        gt.append((1 - r) * pg[j] + r * pg[j + 1])
        est.append(p)
    est, gt = np.array(est), np.array(gt)
    mg, me = gt.mean(0), est.mean(0)
    cov = (gt - mg).T @ (est - me)
    U, _, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    t = mg - R @ me
    res = (est @ R.T + t) - gt
    return float(np.sqrt((res ** 2).sum(1).mean())), len(est)


@pytest.fixture(scope="module")
def seq_mod(vsl):
    return importlib.import_module("visual_slam_amd.synth_sequence")


def test_harness_host_pieces(tmp_path, seq_mod):
    exe = tmp_path / "harness_host_test"
    subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Werror", "-I", str(ROOT / "include"),
                    str(ROOT / "tests" / "cpp" / "harness_host_test.cpp"), "-o", str(exe)], check=True)
    d = tmp_path / "files"
    d.mkdir()
    seq_mod.write_calibration(d / "calib.json")
    # EuRoC csv layout: comment line, CRLF, 19-digit stamps; a short line that must be skipped
    seq = d / "seq"
    for c in (0, 1):
        (seq / ("cam%d" % c) / "data").mkdir(parents=True)
    stamps = [1403715273262142976, 1403715273312143104, 1403715273362142976]
    (seq / "cam0" / "data.csv").write_bytes(b"#timestamp [ns],filename\r\n" + b"short\r\n" +
                                            b"".join(b"%d,%d.png\r\n" % (s, s) for s in stamps))
    (seq / "state_groundtruth_estimate0").mkdir()
    gt_lines = ["#timestamp, p_RS_R_x [m], ...\r\n"]
    for k in range(4):
        gt_lines.append("%d,%.3f,%.3f,%.3f,1,0,0,0,0,0,0,0,0,0,0,0,0\r\n" % (stamps[0] + k * 5000000, 0.5 * k, 0.125 * k, -k))
    (seq / "state_groundtruth_estimate0" / "data.csv").write_text("".join(gt_lines), newline="")

    rng = np.random.default_rng(3)
    img = (rng.integers(0, 256, (61, 97)) // 32 * 32).astype(np.uint8)   # compressible, non-trivial
    img[10:30, 20:70] = rng.integers(0, 256, (20, 50))
    seq_mod.write_pgm(d / "img.pgm", img)
    seq_mod.write_png(d / "img_l0.png", img, level=0)                     # stored blocks
    seq_mod.write_png(d / "img_l6.png", img, level=6)                     # dynamic Huffman
    seq_mod.write_png(d / "img_l9.png", img, level=9)
    seq_mod.write_png(d / "img_filters.png", img, level=6, filters="mixed")
    seq_mod.write_png(d / "img_rgb.png", np.stack([img] * 3, -1), level=6, filters="mixed")
    seq_mod.write_png(d / "img_multi_idat.png", img, level=6, idat_split=101)
    full = (d / "img_l6.png").read_bytes()
    (d / "truncated.png").write_bytes(full[:len(full) // 2])
    # a constant image compresses with zlib's fixed-Huffman block type
    # trajectory alignment fixture
    n_gt = 400
    tg = stamps[0] + np.arange(n_gt, dtype=np.int64) * 5_000_000
    s = np.linspace(0, 3, n_gt)
    pg = np.stack([np.cos(s), 0.3 * s, np.sin(2 * s)], 1)
    te = tg[5:-5:7] + 1_234_567
    pe_true = np.stack([np.interp(te, tg, pg[:, k]) for k in range(3)], 1)
    A = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    if np.linalg.det(A) < 0:
        A[:, 0] *= -1
    pe = (pe_true + 0.01 * rng.normal(size=pe_true.shape)) @ A.T + np.array([3.0, -2.0, 0.5])
    # plus one estimate before the first and one after the last ground-truth stamp (both must be skipped)
    te = np.concatenate([[tg[0] - 10], te, [tg[-1] + 10]])
    pe = np.concatenate([[[9.0, 9.0, 9.0]], pe, [[-9.0, 9.0, 9.0]]])
    expect, n_assoc = _numpy_ate(te, pe, tg, pg)
    assert n_assoc == len(te) - 2
    with open(d / "ate.txt", "w") as f:
        f.write("%d %d %.15e\n" % (len(te), n_gt, expect))
        for t, p in zip(te, pe):
            f.write("%d %.15e %.15e %.15e\n" % (t, *p))
        for t, p in zip(tg, pg):
            f.write("%d %.15e %.15e %.15e\n" % (t, *p))
    r = subprocess.run([str(exe), str(d)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "harness host tests OK" in r.stdout


def test_headless_app_compiles():
    # host program above the C ABI: must build with plain g++ against include/ (link step needs the .so)
    subprocess.run(["g++", "-O0", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", str(ROOT / "include"),
                    str(ROOT / "visual-slam_amd" / "apps" / "slam_headless.cpp")], check=True)


def test_renderer_is_consistent_with_the_camera_model(seq_mod):
    # a world point on a wall must land, through the oracle-independent numpy projection, on the pixel whose
    # ray hit it: unproject -> intersect -> project round trip of the double-sphere model
    intr = seq_mod.CALIB["intrinsics"][0]
    b = seq_mod.ds_unproject_grid(intr)
    assert abs(np.linalg.norm(b, axis=-1) - 1).max() < 1e-12
    fx, fy, cx, cy, xi, al = (intr[k] for k in ("fx", "fy", "cx", "cy", "p1", "p2"))
    p = b * 3.7
    d1 = np.linalg.norm(p, axis=-1)
    d2 = np.sqrt(p[..., 0] ** 2 + p[..., 1] ** 2 + (xi * d1 + p[..., 2]) ** 2)
    den = al * d2 + (1 - al) * (xi * d1 + p[..., 2])
    u, v = fx * p[..., 0] / den + cx, fy * p[..., 1] / den + cy
    uu, vv = np.meshgrid(np.arange(seq_mod.W), np.arange(seq_mod.H))
    assert abs(u - uu).max() < 1e-9 and abs(v - vv).max() < 1e-9
