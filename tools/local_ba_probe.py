"""Local BA (BASELINE configs[2]: 7 keyframes, ~15k landmarks) timing probe: ms per LM iteration and the device
stage times, for the default kernels and for the diagnostic variants given on the command line (name=value ...)."""
import importlib
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry  # noqa: E402

vsl = entry.load_package()
import os  # noqa: E402
if os.environ.get("VSL_SO"):  # a second build of the library (experimental kernel) for same-box comparisons
    vsl._SO = Path(os.environ["VSL_SO"]).resolve()
orc = entry.load_oracle()
synth = importlib.import_module("visual_slam_amd.synth")
ctx = vsl.Context(0)
n_kf = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 7
d = synth.ba_problem(4, n_kf=n_kf, n_lms=20000)
mk = lambda: orc.BaArrays(d["poses"], d["cam_fixed"], d["cam_intr"], d["intr"], d["points"], d["obs_cam"], d["obs_lm"],  # noqa: E731
                          d["obs_uv"], d["cam_model"])
variants = [()] + [tuple(a.split("=")) for a in sys.argv[1:] if "=" in a]
for var in variants:
    if var:
        ctx.set_diagnostic(var[0], int(var[1]))
    ctx.bundle_adjust(mk(), max_iters=2)
    best = None
    for _ in range(5):
        a = mk()
        ctx.synchronize()
        t0 = time.perf_counter()
        s = ctx.bundle_adjust(a, max_iters=20)
        ms = 1e3 * (time.perf_counter() - t0)
        if best is None or ms < best[0]:
            best = (ms, s)
    ms, s = best
    ctx.set_profiling(1)   # stage times from a separate, profiled run (the stage events slow the loop down)
    sp = ctx.bundle_adjust(mk(), max_iters=20)
    ctx.set_profiling(0)
    s.linearize_ms, s.schur_ms, s.solve_ms = sp.linearize_ms, sp.schur_ms, sp.solve_ms
    print("%-24s iters %d  %.3f ms/iter (total %.2f ms)  device: linearize %.3f schur %.3f solve %.3f ms  final cost %.9e"
          % (var or "default", s.iterations, ms / s.iterations, ms, s.linearize_ms, s.schur_ms, s.solve_ms, s.final_cost), flush=True)
    if var:
        ctx.set_diagnostic(var[0], 0)
