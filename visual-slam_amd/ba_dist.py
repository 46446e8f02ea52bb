"""Multi-GPU global bundle adjustment: the Levenberg-Marquardt loop over vsl_ba_session, one process
per GPU, collectives through torch.distributed (backend "nccl" = RCCL over xGMI).

Replaces, for maps too large for one solve to be interactive (~500 keyframes / ~1e5 landmarks,
BASELINE.json configs[4]), what the reference hands to one ceres::Solve in
global_bundle_adjustment (include/visnav/loop_closure_utils.h:672-748).  The LM policy is the same
[upstream] Ceres policy as visual-slam_amd/csrc/ba.hip::vsl_bundle_adjust and oracle/orc_ba.cpp.

Per LM iteration: ONE SUM all-reduce of the packed partial reduced camera system
[S | rhs | diag H | g_c | cost] (n*n + 3n + 2 doubles, n = 6 x free cameras) and ONE SUM all-reduce
of 8 scalars; after an accepted step one MAX all-reduce of a scalar.  Every rank factorises the same
system redundantly: RCCL leaves identical bytes on every rank, so all ranks take the same branch.
With the "gloo" backend (CPU tests, or several ranks sharing one GPU) the buffers hop through host
memory; with "nccl" they stay in HBM.
"""
import math
from types import SimpleNamespace

import numpy as np


def _allreduce(t, op_name, group=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    op = dist.ReduceOp.SUM if op_name == "sum" else dist.ReduceOp.MAX
    if dist.get_backend(group) == "gloo" and t.is_cuda:
        c = t.cpu()
        dist.all_reduce(c, op=op, group=group)
        t.copy_(c)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def bundle_adjust_distributed(pkg, ctx, arr, use_huber=True, huber=1.0, max_iters=20, verbosity=0, group=None):
    """arr: flattened problem (same object as Context.bundle_adjust takes), identical on every rank.
    Optimises arr.poses / arr.points in place on every rank; returns a summary namespace.

    Stream contract: the session kernels run on ctx's stream, the buffer fills / element-wise math / collectives are
    torch work.  The whole loop therefore runs with ctx's stream as torch's CURRENT stream (an ExternalStream view of
    it), so that both kinds of work are ordered on one stream and an RCCL collective (which torch orders against the
    current stream on both sides) can never overtake or be overtaken by a session kernel."""
    import torch
    handle = ctx.stream()  # None = the device's default (null) stream
    stream = torch.cuda.ExternalStream(handle) if handle else torch.cuda.default_stream()
    with torch.cuda.stream(stream):
        return _bundle_adjust_distributed(pkg, ctx, arr, use_huber, huber, max_iters, verbosity, group)


def _bundle_adjust_distributed(pkg, ctx, arr, use_huber, huber, max_iters, verbosity, group):
    import torch
    import torch.distributed as dist
    from . import dist as vdist

    distributed = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    counts = np.bincount(arr.obs_lm, minlength=len(arr.points))
    ranges = vdist.landmark_ranges(counts, world)
    first, cnt = ranges[rank]
    if cnt < 1:
        raise ValueError("rank %d owns no landmarks (%d landmarks over %d ranks)" % (rank, len(arr.points), world))
    sess = pkg.BaSession(ctx, arr, use_huber, huber, first, cnt)
    n = sess.n
    dev = torch.device("cuda", torch.cuda.current_device())
    f64 = torch.float64
    bufA = torch.zeros(n + 1, dtype=f64, device=dev)
    packB = torch.zeros(n * n + 3 * n + 2, dtype=f64, device=dev)
    packC = torch.zeros(8, dtype=f64, device=dev)
    gl = torch.zeros(1, dtype=f64, device=dev)

    # iteration 0: cost, Jacobi scaling from the global column norms
    sess.linearize(0)
    sess.hdiag_cost(bufA.data_ptr())
    ctx.synchronize()
    _allreduce(bufA, "sum", group)
    scale_c = 1.0 / (1.0 + torch.sqrt(bufA[:n]))
    sess.set_scale(bufA.data_ptr())
    summary = SimpleNamespace(initial_cost=float(bufA[n].item()), final_cost=0.0, iterations=0, successful_steps=0,
                              termination=0, world=world)
    radius, decrease, it, invalid, refresh = 1e4, 2.0, 0, 0, 1
    cost, gmax = summary.initial_cost, math.inf
    while True:
        sess.reduce(radius, packB.data_ptr(), gl.data_ptr())
        ctx.synchronize()
        _allreduce(packB, "sum", group)
        if refresh:
            _allreduce(gl, "max", group)
            cost = float(packB[n * n + 3 * n].item())
            g_c = packB[n * n + 2 * n:n * n + 3 * n]
            gmax = max(float((g_c / scale_c).abs().max().item()) if n else 0.0, float(gl.item()))
        if it >= max_iters:
            summary.termination = 0
            break
        if gmax <= 1e-10:
            summary.termination = 2
            break
        if radius <= 1e-32:
            summary.termination = 4
            break
        it += 1
        sess.step(packB.data_ptr(), radius, refresh, packC.data_ptr())
        ctx.synchronize()
        _allreduce(packC, "sum", group)
        c = packC.tolist()
        cams_step2, cams_x2 = c[5] / world, c[6] / world
        ok = c[0] == 0.0 and c[1] > 0.0
        if not ok:
            invalid += 1
            if invalid >= 5:
                summary.termination = 4
                break
            radius *= 0.5
            refresh = 0
            continue
        invalid = 0
        step_norm = math.sqrt(max(c[2] - (world - 1) * cams_step2, 0.0))
        x_norm = math.sqrt(max(c[3] - (world - 1) * cams_x2, 0.0))
        if step_norm <= 1e-8 * (x_norm + 1e-8):
            summary.termination = 3
            break
        cost_change = cost - c[4]
        if abs(cost_change) <= 1e-6 * cost:
            summary.termination = 1
            break
        rel = cost_change / c[1]
        if verbosity >= 2 and rank == 0:
            print("%4d % .6e % .3e % .3e % .3e % .3e % .3e" % (it, c[4], cost_change, gmax, step_norm, rel, radius))
        if rel > 1e-3:
            sess.accept()
            sess.linearize(1)
            refresh = 1
            summary.successful_steps += 1
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rel - 1.0) ** 3))
            decrease = 2.0
        else:
            radius /= decrease
            decrease *= 2.0
            refresh = 0
    summary.iterations = it
    summary.final_cost = cost
    poses, pts_own = sess.download()
    arr.poses[:] = poses
    if world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, (first, pts_own), group=group)
        for f, p in parts:
            arr.points[f:f + len(p)] = p
    else:
        arr.points[first:first + cnt] = pts_own
    sess.close()
    return summary
