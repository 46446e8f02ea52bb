"""Renders a looping EuRoC-layout sequence and runs slam_headless on it with the relocalisation / loop-closure
branches on: python tools/loop_probe.py [n_frames] [radius] [step] -- extra flags go to slam_headless."""
import importlib
import json
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry  # noqa: E402

entry.load_package()
sq = importlib.import_module("visual_slam_amd.synth_sequence")
synth = importlib.import_module("visual_slam_amd.synth")
args = sys.argv[1:]
extra = []
if "--" in args:
    extra = args[args.index("--") + 1:]
    args = args[:args.index("--")]
n = int(args[0]) if args else 200
radius = float(args[1]) if len(args) > 1 else 1.2
step = float(args[2]) if len(args) > 2 else 0.045
with tempfile.TemporaryDirectory(prefix="vsl_loop_") as d:
    t = time.time()
    sq.render_sequence(d, n_frames=n, seed=1, step_m=step, radius=radius, workers=12)
    print("rendered %d frames in %.1f s (loop every %.0f frames)" % (n, time.time() - t, 2 * 3.14159265 * radius / step), flush=True)
    voc = Path(d) / "voc.txt"
    voc.write_text(synth.vocabulary_text(3, 10, 4))
    for flags in ([], ["--voc-path", str(voc)] + extra):
        r = subprocess.run([str(ROOT / "visual-slam_amd" / "slam_headless"), "--dataset-path", d, "--cam-calib", d + "/calib.json"] + flags,
                           capture_output=True, text=True, timeout=900, env=dict(__import__("os").environ, VISNAV_AMD_TRACE="1"))
        print(" ".join(flags) or "(vo only)")
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-2000:], flush=True)
        if "--trace" in flags:
            print("\n".join(l for l in r.stderr.splitlines() if l.lstrip().startswith(("reloc", "loop", "pnp")))[-5000:])
