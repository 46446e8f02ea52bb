#!/usr/bin/env python3
"""Global bundle adjustment (BASELINE.json configs[4]: ~500 keyframes, ~100k landmarks) through the
multi-GPU path.  Run under torchrun for N > 1 (backend nccl); prints timings on rank 0.

    python tools/global_ba_bench.py [--kf 500] [--lms 100000] [--iters 5]
"""
import argparse
import importlib
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kf", type=int, default=500)
    ap.add_argument("--lms", type=int, default=100000)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--radius", type=float, default=200.0)
    ap.add_argument("--dense", action="store_true", help="force the dense reduced camera system (diagnostic)")
    ap.add_argument("--single-call", action="store_true", help="vsl_bundle_adjust instead of the session path (one rank)")
    args = ap.parse_args()
    import torch
    vsl = entry.load_package()
    synth = importlib.import_module("visual_slam_amd.synth")
    vdist = importlib.import_module("visual_slam_amd.dist")
    ba_dist = importlib.import_module("visual_slam_amd.ba_dist")
    rank, world, local_rank = vdist.env_rank_world()
    torch.cuda.set_device(local_rank)
    vdist.init("nccl")
    t0 = time.perf_counter()
    d = synth.ba_problem(5, n_kf=args.kf, n_lms=args.lms, loop_radius=args.radius, max_range=15.0)
    gen_s = time.perf_counter() - t0

    class A:
        pass
    arr = A()
    for k in ("poses", "cam_fixed", "cam_intr", "intr", "points", "obs_cam", "obs_lm", "obs_uv"):
        setattr(arr, k, np.ascontiguousarray(d[k]))
    arr.obs_uv = np.ascontiguousarray(arr.obs_uv, np.float64)
    arr.cam_model = d["cam_model"]
    ctx = vsl.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    if args.dense:
        ctx.set_diagnostic("ba_force_dense", 1)
    import copy
    run = (lambda a, it, v: ctx.bundle_adjust(a, max_iters=it, verbosity=v)) if args.single_call else \
        (lambda a, it, v: ba_dist.bundle_adjust_distributed(vsl, ctx, a, max_iters=it, verbosity=v))
    times = {}
    for it in (1, 2, args.iters, 2, args.iters):      # first call = warm-up; marginal time per iteration from two lengths
        a = copy.deepcopy(arr)
        vdist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s = run(a, it, 1 if it == args.iters else 0)
        torch.cuda.synchronize()
        vdist.barrier()
        dt = time.perf_counter() - t0
        times[it] = min(dt, times.get(it, (1e9,))[0]), s
    dt, s = times[args.iters]
    if rank == 0 and args.iters > 2:
        print("marginal ms per LM iteration: %.2f" % (1e3 * (times[args.iters][0] - times[2][0]) / max(times[args.iters][1].iterations - times[2][1].iterations, 1)))
    if rank == 0:
        n = 6 * int((arr.cam_fixed == 0).sum())
        print("global BA: %d cameras (%d x %d reduced system), %d landmarks, %d observations, %d rank(s)" %
              (len(arr.poses), n, n, len(arr.points), len(arr.obs_cam), world))
        print("generation %.1f s; solve %.3f s for %d LM iterations = %.1f ms/iter; cost %.6e -> %.6e" %
              (gen_s, dt, s.iterations, 1e3 * dt / max(s.iterations, 1), s.initial_cost, s.final_cost))
    ctx.close()


if __name__ == "__main__":
    main()
