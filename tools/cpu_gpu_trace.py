"""Runs the MI355X headless binary and the CPU-oracle one on the same rendered sequence with --trace and prints the
first frames where their per-frame counters differ."""
import importlib, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry
entry.load_package()
sq = importlib.import_module("visual_slam_amd.synth_sequence")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
with tempfile.TemporaryDirectory() as d:
    sq.render_sequence(d, n_frames=n, seed=1, step_m=0.04, radius=1.6, workers=8)
    outs = {}
    for name, exe in (("gpu", ROOT / "visual-slam_amd" / "slam_headless"), ("cpu", ROOT / "oracle" / "_cpu" / "slam_headless_cpu")):
        r = subprocess.run([str(exe), "--dataset-path", d, "--cam-calib", d + "/calib.json", "--trace"] + sys.argv[2:], capture_output=True, text=True)
        outs[name] = [l for l in r.stderr.splitlines() if l.startswith("frame")]
    shown = 0
    for a, b in zip(outs["gpu"], outs["cpu"]):
        if a != b and shown < 6:
            print("GPU", a)
            print("CPU", b)
            shown += 1
    print("lines", len(outs["gpu"]), len(outs["cpu"]), "differing", sum(a != b for a, b in zip(outs["gpu"], outs["cpu"])))
