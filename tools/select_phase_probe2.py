import importlib, sys, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
synth = importlib.import_module("visual_slam_amd.synth")
ctx = pkg.Context(0)
B = 128
fr = pkg.Frames(ctx, 2 * B, 752, 480, 1500, max_pairs=B)
pairs = [synth.stereo_pair(s) for s in range(100, 108)]
batch = np.stack([pairs[(k // 2) % 8][k % 2] for k in range(2 * B)])
fr.upload(0, batch)
for nf in (1500, 600, 100, 10):
    ctx.set_profiling(True)
    for it in range(8):
        if it == 3:
            ctx.reset_profiling()
        fr.detect_describe(0, 2 * B, nf, True)
    ctx.synchronize()
    st = ctx.stage_ms()
    print(nf, {k: round(v[0] / max(v[1], 1), 4) for k, v in st.items()})
