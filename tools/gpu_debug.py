#!/usr/bin/env python3
"""Ad-hoc GPU-vs-oracle diagnostics (prints mismatch statistics instead of asserting)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry
import importlib

vsl = entry.load_package(); orc = entry.load_oracle()
synth = importlib.import_module("visual_slam_amd.synth")
ctx = vsl.Context(0)
rng = np.random.default_rng(3)
left, right = synth.stereo_pair(11)
imgs = dict(left=left, noise=rng.integers(0, 256, (480, 752), dtype=np.uint8),
            small=rng.integers(0, 256, (97, 131), dtype=np.uint8),
            flat=np.full((480, 752), 90, np.uint8))
for name, img in imgs.items():
    got = ctx.min_eig_response(img); exp = orc.min_eig_response(img)
    bad = np.argwhere(got.view(np.uint32) != exp.view(np.uint32))
    print(name, "response mismatches:", len(bad), "of", got.size)
    for y, x in bad[:8]:
        print("   (x=%d,y=%d) got %r exp %r" % (x, y, got[y, x], exp[y, x]))
    if len(bad):
        print("   rows", np.unique(bad[:, 0])[:20], "cols", np.unique(bad[:, 1])[:20])
    for nf in (1500, 100):
        xy, ang, desc = ctx.detect_describe(img, nf, True)
        oxy, oang, odesc = orc.detect_describe(img, nf, True)
        same_xy = len(xy) == len(oxy) and np.array_equal(xy, oxy)
        print("  nf", nf, "n", len(xy), len(oxy), "xy equal", same_xy)
        if not same_xy:
            k = min(len(xy), len(oxy))
            d = np.nonzero((xy[:k] != oxy[:k]).any(1))[0]
            print("    first diffs at", d[:10], xy[d[:3]], oxy[d[:3]])
        else:
            print("    angle equal", np.array_equal(ang.view(np.uint64), oang.view(np.uint64)),
                  "desc equal", np.array_equal(desc, odesc), "desc rows differing", int((desc != odesc).any(1).sum()))
            if not np.array_equal(ang.view(np.uint64), oang.view(np.uint64)):
                d = np.nonzero(ang != oang)[0]; print("    angle diffs", len(d), ang[d[:3]], oang[d[:3]])
