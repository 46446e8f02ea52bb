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

if os.environ.get("MX_TIMING"):
    import ctypes
    out = (ctypes.c_longlong * 64)()
    vsl.load().vsl_mx_timing(out)
    for w in range(12):
        a, b_, c_, n = out[4 * w:4 * w + 4]
        print("wave %2d: per step (24 steps)  compute %6.0f  fill %6.0f  barrier %6.0f ticks;  whole loop %d ticks = %d wall-clock ticks (100 MHz) -> %.2f GHz"
              % (w, a / 24, b_ / 24, c_ / 24, a + b_ + c_, n, (a + b_ + c_) / max(n, 1) * 0.1))

    tl = (ctypes.c_longlong * (4 * 3072))()
    vsl.load().vsl_mx_timeline(tl)
    tl = np.array(tl[:], np.int64).reshape(3072, 4)
    t0 = tl[:, 0].min()
    ent, ls, le = (tl[:, 0] - t0) / 100.0, (tl[:, 1] - t0) / 100.0, (tl[:, 2] - t0) / 100.0
    print("forward launch timeline (us from the first entry): last exit %.1f" % le.max())
    for q in (0, 767, 768, 1000, 1535, 1536, 2304, 3071):
        print("  block %4d: entry %6.1f  loop %6.1f .. %6.1f   (prologue %.1f us, loop %.1f us)  hw_id %08x" % (q, ent[q], ls[q], le[q], ls[q] - ent[q], le[q] - ls[q], tl[q, 3]))
    print("  prologue us: mean %.1f  p10 %.1f  p90 %.1f;  loop us: mean %.1f p10 %.1f p90 %.1f" % ((ls - ent).mean(), np.percentile(ls - ent, 10), np.percentile(ls - ent, 90), (le - ls).mean(), np.percentile(le - ls, 10), np.percentile(le - ls, 90)))
    hist, _ = np.histogram(ent, bins=16, range=(0, le.max()))
    print("  entries per %.1f us bin:" % (le.max() / 16), hist.tolist())
