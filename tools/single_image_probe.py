"""One-image detect + describe latency (the tracking loop's regime): ms per call over 300 calls, device stage times."""
import importlib
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry  # noqa: E402

vsl = entry.load_package()
synth = importlib.import_module("visual_slam_amd.synth")
ctx = vsl.Context(0)
left, right = synth.stereo_pair(7)
for _ in range(20):
    ctx.detect_describe(left, 1500)
ctx.synchronize()
t0 = time.perf_counter()
N = 300
for _ in range(N):
    xy, ang, d = ctx.detect_describe(left, 1500)
dt = (time.perf_counter() - t0) / N
ctx.reset_profiling()
ctx.set_profiling(1)
for _ in range(50):
    ctx.detect_describe(left, 1500)
st = ctx.stage_ms()
ctx.set_profiling(0)
print("one image: %.1f us per detect_describe call, %d keypoints; device us per call: %s"
      % (1e6 * dt, len(xy), {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in st.items() if v[1] > 0}))
