"""Device time of ONE matchDescriptors call (1500 x 1500, the per-keyframe stereo match / relocalisation shape):
launches of fewer than 8 pairs run both directions in one launch of the FP4 kernel.  VSL_SO selects another build."""
import importlib
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry  # noqa: E402

vsl = entry.load_package()
if os.environ.get("VSL_SO"):
    vsl._SO = Path(os.environ["VSL_SO"]).resolve()
synth = importlib.import_module("visual_slam_amd.synth")
ctx = vsl.Context(0)
left, right = synth.stereo_pair(7)
_, _, d1 = ctx.detect_describe(left, 1500)
_, _, d2 = ctx.detect_describe(right, 1500)
for _ in range(50):
    m = ctx.match_descriptors(d1, d2, 70, 1.2)
ctx.set_profiling(True)
ctx.reset_profiling()
for _ in range(200):
    m = ctx.match_descriptors(d1, d2, 70, 1.2)
ms, n = ctx.stage_ms()["match"]
print("single pair %d x %d: %.1f us of device time per call (%d matches)" % (len(d1), len(d2), 1e3 * ms / n, len(m)))
ctx.close()
