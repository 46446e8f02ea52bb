#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer (literal drop-in) entry points: one image per call,
upload + kernels + download, synchronous -- what src/slam.cpp would see after the two-line patch."""
import importlib
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry  # noqa: E402

vsl = entry.load_package()
synth = importlib.import_module("visual_slam_amd.synth")
ctx = vsl.Context(0)
pairs = [synth.stereo_pair(100 + i) for i in range(4)]
for _ in range(3):
    ctx.detect_describe(pairs[0][0], 1500, True)
n = 100
t0 = time.perf_counter()
for i in range(n):
    l, r = pairs[i % 4]
    _, _, d1 = ctx.detect_describe(l, 1500, True)
    _, _, d2 = ctx.detect_describe(r, 1500, True)
    ctx.match_descriptors(d1, d2, 70, 1.2)
dt = time.perf_counter() - t0
print("host-buffer API: %.1f stereo frames/s (%.3f ms per stereo frame: 2x detect_describe + match, "
      "incl. PCIe and ctypes)" % (n / dt, 1e3 * dt / n))
