"""Per-stage HIP-event times of one bench.py launch (512 stereo frames = 1024 images of 1500 keypoints, one stream):
response, select, describe, match.  name=value arguments set diagnostic knobs first.  Also prints a checksum of the
outputs so that two builds can be compared."""
import importlib
import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry  # noqa: E402

vsl = entry.load_package()
import os  # noqa: E402
if os.environ.get("VSL_SO"):  # a second build of the library (old / experimental kernel) for same-box comparisons
    vsl._SO = Path(os.environ["VSL_SO"]).resolve()
synth = importlib.import_module("visual_slam_amd.synth")
Bu = int(os.environ.get("BU", "512"))  # stereo frames per launch
base = np.concatenate([synth.stereo_pair_variants(10 + s, 4, margin=24) for s in range(16)])  # 64 distinct pairs
imgs = np.concatenate([base] * max(1, Bu // len(base)))[:Bu].reshape(2 * Bu, 480, 752)
ctx = vsl.Context(0)
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=")
        ctx.set_diagnostic(k, int(v))
fr = vsl.Frames(ctx, 2 * Bu, 752, 480, 1500, max_pairs=Bu)
fr.upload(0, imgs)
pairs = np.array([[2 * k, 2 * k + 1] for k in range(Bu)], np.int32)


def run():
    fr.detect_describe(0, 2 * Bu, 1500, True)
    fr.resolve_ties()
    fr.match(pairs, 70, 1.2)


for _ in range(25):
    run()
ctx.synchronize()
ctx.set_profiling(True)
ctx.reset_profiling()
for _ in range(10):
    run()
ctx.synchronize()
st = {k: ms / n for k, (ms, n) in ctx.stage_ms().items() if n}
nk, nm = fr.counts(2 * Bu, Bu)
chk = 0
for s in (0, 1, 77 % (2 * Bu), 500 % (2 * Bu)):
    xy, ang, d = fr.keypoints(s)
    chk = zlib.crc32(xy.tobytes() + d.tobytes(), chk)
chk = zlib.crc32(fr.matches(min(5, Bu - 1)).tobytes() + nm.tobytes(), chk)
print("stages ms/launch:", {k: round(v, 4) for k, v in st.items()}, "sum %.4f" % sum(st.values()),
      "| candidates/img %.0f kp %.1f matches %.1f crc %08x" % (fr.candidate_counts(2 * Bu).mean(), nk.mean(), nm.mean(), chk), flush=True)
