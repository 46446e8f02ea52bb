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
for i in range(300):  # warm-up: code objects of every kernel, scratch buffers, and the chip's clock ramp (~50 ms)
    _, _, w1 = ctx.detect_describe(pairs[i % 4][0], 1500, True)
    _, _, w2 = ctx.detect_describe(pairs[i % 4][1], 1500, True)
    ctx.match_descriptors(w1, w2, 70, 1.2)
n = 200
t0 = time.perf_counter()
for i in range(n):
    l, r = pairs[i % 4]
    _, _, d1 = ctx.detect_describe(l, 1500, True)
    _, _, d2 = ctx.detect_describe(r, 1500, True)
    ctx.match_descriptors(d1, d2, 70, 1.2)
dt = time.perf_counter() - t0
print("host-buffer API: %.1f stereo frames/s (%.3f ms per stereo frame: 2x detect_describe + match, "
      "incl. PCIe and ctypes)" % (n / dt, 1e3 * dt / n))
# where the time goes: device time per stage (HIP events) and the wall time of each call type
ctx.set_profiling(True)
ctx.reset_profiling()
t_dd = t_m = 0.0
for i in range(n):
    l, r = pairs[i % 4]
    t0 = time.perf_counter()
    _, _, d1 = ctx.detect_describe(l, 1500, True)
    _, _, d2 = ctx.detect_describe(r, 1500, True)
    t1 = time.perf_counter()
    ctx.match_descriptors(d1, d2, 70, 1.2)
    t_m += time.perf_counter() - t1
    t_dd += t1 - t0
st = ctx.stage_ms()
print("wall per call: detect_describe %.3f ms, match_descriptors %.3f ms" % (1e3 * t_dd / (2 * n), 1e3 * t_m / n))
print("device ms per call:", {k: round(v[0] / max(v[1], 1), 4) for k, v in st.items() if v[1]})
