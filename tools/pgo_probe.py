"""Pose graph optimisation timing (loop_closure_utils.h:446-587 on the device, visual-slam_amd/csrc/pgo.hip): ms per LM
iteration for keyframe graphs of the reference's shape (odometry + covisibility window + loop edge: cyclic band form) and
for the same graphs with the dense solver (ba_force_dense)."""
import importlib
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry  # noqa: E402

vsl = entry.load_package()
orc = entry.load_oracle()
synth = importlib.import_module("visual_slam_amd.synth")
ctx = vsl.Context(0)
for n, w in ((100, 6), (500, 6), (500, 12)):
    d = synth.pose_graph(5, n, 0, meas_noise=0.002, drift=0.02, window=w)
    mk = lambda: orc.PgoArrays(d["poses"], d["node_fixed"], d["edge_a"], d["edge_b"], d["edge_meas"])  # noqa: E731
    for dense in (0, 1):
        ctx.set_diagnostic("ba_force_dense", dense)
        ctx.pose_graph_optimize(mk(), True, 1.0, 3)
        best = None
        for _ in range(3):
            a = mk()
            ctx.synchronize()
            t0 = time.perf_counter()
            s = ctx.pose_graph_optimize(a, True, 1.0, 20)
            ms = 1e3 * (time.perf_counter() - t0)
            best = ms if best is None else min(best, ms)
        print("%4d nodes, window %2d, %5d edges, %s: %2d iterations, %.2f ms total, %.3f ms per iteration, final cost %.6e"
              % (n, w, len(d["edge_a"]), "dense" if dense else "band ", s.iterations, best, best / max(s.iterations, 1), s.final_cost), flush=True)
    ctx.set_diagnostic("ba_force_dense", 0)
