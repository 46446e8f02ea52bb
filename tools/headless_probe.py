"""Developer probe: render a synthetic EuRoC-layout sequence and run slam_headless on it with the given extra flags."""
import importlib
import pathlib
import subprocess
import sys
import tempfile
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as g

g.load_package()
sq = importlib.import_module("visual_slam_amd.synth_sequence")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
d = tempfile.mkdtemp(prefix="seq")
t = time.time()
sq.render_sequence(d, n_frames=n, seed=1, step_m=0.04, radius=1.6)
print("rendered %d frames in %.1f s" % (n, time.time() - t), flush=True)
exe = ROOT / "visual-slam_amd" / "slam_headless"
for extra in [a.split() for a in (sys.argv[2:] or [""])]:
    r = subprocess.run([str(exe), "--dataset-path", d, "--cam-calib", d + "/calib.json", *extra], capture_output=True, text=True)
    print(extra, r.returncode, r.stdout.strip()[-1500:], r.stderr.strip()[-800:], flush=True)
