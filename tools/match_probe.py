"""Matcher timing probe: 512 stereo pairs (bench.py's launch size) of 1500-keypoint frames, HIP-event stage times of
the matrix-core matcher for the default kernel and for the diagnostic variants given as name=value arguments."""
import importlib
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry  # noqa: E402

vsl = entry.load_package()
import os  # noqa: E402
if os.environ.get("VSL_SO"):  # a second build of the library (experimental kernel) for same-box comparisons
    vsl._SO = Path(os.environ["VSL_SO"]).resolve()
synth = importlib.import_module("visual_slam_amd.synth")
Bu = 512
base = np.concatenate([synth.stereo_pair_variants(100 + s, 4, margin=24) for s in range(16)])  # 64 distinct pairs
imgs = np.concatenate([base] * (Bu // len(base))).reshape(2 * Bu, 480, 752)
ctx = vsl.Context(0)
fr = vsl.Frames(ctx, 2 * Bu, 752, 480, 1500, max_pairs=Bu)
fr.upload(0, imgs)
fr.detect_describe(0, 2 * Bu, 1500, True)
fr.resolve_ties()
pairs = np.array([[2 * k, 2 * k + 1] for k in range(Bu)], np.int32)
ref = None
knobs = [tuple(a.split("=")) for a in sys.argv[1:] if "=" in a]
for var in [()] + ([tuple(knobs)] if knobs else []) + [()]:
    for kv in var:
        ctx.set_diagnostic(kv[0], int(kv[1]))
    for _ in range(30):
        fr.match(pairs, 70, 1.2)
    ctx.synchronize()
    ctx.set_profiling(True)
    ctx.reset_profiling()
    for _ in range(20):
        fr.match(pairs, 70, 1.2)
    ctx.synchronize()
    ms, n = ctx.stage_ms()["match"]
    ctx.set_profiling(False)
    nk, nm = fr.counts(2 * Bu, Bu)
    m0 = fr.matches(3)
    if ref is None:
        ref = (nm.copy(), m0.copy())
    same = np.array_equal(ref[0], nm) and np.array_equal(ref[1], m0)
    macs = sum(2 * int(nk[2 * k]) * int(nk[2 * k + 1]) * 256 for k in range(Bu))
    print("%-26s match %.4f ms per launch  %.0f TOP/s (%.3f of 5000)  same results %s  mean kp %.0f"
          % (var or "default", ms / n, 2 * macs / (ms / n * 1e-3) / 1e12, 2 * macs / (ms / n * 1e-3) / 1e12 / 5000, same, nk.mean()), flush=True)
    for kv in var:
        ctx.set_diagnostic(kv[0], 0)
