"""Developer probe: one 128-frame batch on one stream vs two 64-frame half-batches on two streams."""
import importlib
import pathlib
import sys
import time

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import __graft_entry__ as g

pkg = g.load_package()
import torch

synth = importlib.import_module("visual_slam_amd.synth")
pairs = [synth.stereo_pair(s) for s in range(100, 108)]


def make(n_streams, B):
    out = []
    for i in range(n_streams):
        st = torch.cuda.Stream()
        ctx = pkg.Context(0, stream=st.cuda_stream)
        b = B // n_streams
        fr = pkg.Frames(ctx, 2 * b, 752, 480, 1500, max_pairs=b)
        batch = np.stack([pairs[(k // 2) % 8][k % 2] for k in range(2 * b)])
        fr.upload(0, batch)
        sp = np.array([[2 * k, 2 * k + 1] for k in range(b)], np.int32)
        out.append((st, ctx, fr, sp, b))
    return out


def run(units, steps):
    for _ in range(steps):
        for st, ctx, fr, sp, b in units:
            fr.detect_describe(0, 2 * b, 1500, True)
        for st, ctx, fr, sp, b in units:
            fr.resolve_ties()
            fr.match(sp, 70, 1.2)
    torch.cuda.synchronize()


for n_streams in (1, 2, 3, 4):
    for B in (128 * n_streams, 256 * n_streams):
        units = make(n_streams, B)
        run(units, 3)
        t0 = time.perf_counter()
        run(units, 20)
        dt = time.perf_counter() - t0
        print("streams %d batch %d: %.1f frames/s" % (n_streams, B, 20 * B / dt), flush=True)
        del units
