import sys, time, importlib
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry
vsl = entry.load_package(); orc = entry.load_oracle()
synth = importlib.import_module("visual_slam_amd.synth")
ctx = vsl.Context(0)
def arrs(d): return orc.BaArrays(d["poses"], d["cam_fixed"], d["cam_intr"], d["intr"], d["points"], d["obs_cam"], d["obs_lm"], d["obs_uv"], d["cam_model"])
t=time.time(); d = synth.ba_problem(61, n_kf=40, n_lms=6000, loop_radius=6.0); print("gen", time.time()-t, "obs", len(d["obs_cam"]), "lms", len(d["points"]), flush=True)
a = arrs(d)
t=time.time(); S,g,c = ctx.ba_linearize(a); print("gpu linearize", time.time()-t, flush=True)
t=time.time(); S,g,c = ctx.ba_linearize(a); print("gpu linearize 2", time.time()-t, flush=True)
t=time.time(); eS,eg,ec = orc.ba_linearize(a); print("cpu linearize", time.time()-t, flush=True)
a1 = arrs(d)
t=time.time(); s = ctx.bundle_adjust(a1, max_iters=6, verbosity=2); print("gpu BA 6 it", time.time()-t, s.linearize_ms, s.schur_ms, s.solve_ms, s.total_ms, flush=True)
a2 = arrs(d)
t=time.time(); s = orc.bundle_adjust(a2, max_iters=6, verbosity=2); print("cpu BA 6 it", time.time()-t, flush=True)
