"""Multi-GPU global bundle adjustment from Python: a thin wrapper over the C++ Levenberg-Marquardt loop
`vsl_global_bundle_adjust` / `vsl_ba_session_solve` (visual-slam_amd/csrc/ba.hip), one process per GPU.

Replaces, for maps too large for one solve to be interactive (~500 keyframes / ~1e5 landmarks, BASELINE.json
configs[4]), what the reference hands to one ceres::Solve in global_bundle_adjustment
(include/visnav/loop_closure_utils.h:672-748).  The loop, the step policy ([upstream] Ceres, the same as
vsl_bundle_adjust and oracle/orc_ba.cpp) and the landmark partition live in the library; what this module adds is the
collective: the library calls back with a DEVICE pointer and a count, and the callback all-reduces it through
torch.distributed -- backend "nccl" (= RCCL over xGMI) directly on a device tensor, backend "gloo" (CPU tests, or
several ranks sharing one GPU) through host memory.  A C++ caller links RCCL instead and passes ncclAllReduce
(include/visnav_amd/bundle_adjustment.h, VISNAV_AMD_WORLD).

Per LM iteration: ONE SUM all-reduce of the packed partial reduced camera system [S | rhs | diag H | g_c | cost]
(band form: n * (bandwidth + 33) + 3n + 66 doubles instead of n * n + 3n + 2), ONE SUM of 8 scalars, and after an
accepted step one MAX of a scalar.  Every rank factorises the same system redundantly: the all-reduce leaves identical
bytes on every rank, so all ranks take the same branch.
"""
import ctypes as C
from types import SimpleNamespace

import numpy as np

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p)


class _DeviceDoubles:
    """`count` float64 values at device address `ptr`, exposed through __cuda_array_interface__ so that
    torch.as_tensor() wraps the memory without copying it."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 3, "strides": None}


def _make_allreduce(pkg, ctx, group):
    """The callback the C++ loop uses for its collectives (device pointer, count of doubles, op, stream)."""
    import torch
    import torch.distributed as dist

    state = {"error": None}

    def fn(_user, buf, count, op, _stream):
        try:
            rop = dist.ReduceOp.SUM if op == 0 else dist.ReduceOp.MAX
            nbytes = 8 * int(count)
            if dist.get_backend(group) == "gloo":
                host = np.empty(int(count), np.float64)
                ctx._ck(ctx.L.vsl_ctx_memcpy(ctx.h, host.ctypes.data_as(C.c_void_p), C.c_void_p(buf), C.c_size_t(nbytes), 1))
                t = torch.from_numpy(host)
                dist.all_reduce(t, op=rop, group=group)
                ctx._ck(ctx.L.vsl_ctx_memcpy(ctx.h, C.c_void_p(buf), host.ctypes.data_as(C.c_void_p), C.c_size_t(nbytes), 0))
            else:
                # RCCL: IN PLACE on the library's own device buffer -- a tensor that aliases it (no allocation, no copies)
                # -- ordered by torch against its current stream, which is the context's stream (see below): the
                # collective waits for the kernels that produced the buffer and the solver's next kernel waits for the
                # collective through stream events; the host never synchronises (round 2 allocated a tensor, copied
                # device-to-device twice and called current_stream().synchronize() per collective)
                t = torch.as_tensor(_DeviceDoubles(int(buf), int(count)), device=torch.device("cuda", torch.cuda.current_device()))
                dist.all_reduce(t, op=rop, group=group)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            state["error"] = e
            return 1

    return ALLREDUCE_FN(fn), state


def bundle_adjust_distributed(pkg, ctx, arr, use_huber=True, huber=1.0, max_iters=20, verbosity=0, group=None,
                              collectives_at_world_one=False, solo=False):
    """arr: flattened problem (same object as Context.bundle_adjust takes), identical on every rank.
    Optimises arr.poses / arr.points in place on every rank; returns a summary namespace.

    Stream contract: the library's kernels run on ctx's stream and torch's collectives are ordered against torch's
    CURRENT stream, so the call runs with ctx's stream as torch's current stream (an ExternalStream view of it)."""
    import torch
    import torch.distributed as dist

    # solo: this rank solves the whole problem by itself, no collective (the world-1 reference of a multi-rank run)
    distributed = dist.is_available() and dist.is_initialized() and not solo
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    handle = ctx.stream()  # None = the device's default (null) stream
    stream = torch.cuda.ExternalStream(handle) if handle else torch.cuda.default_stream()
    with torch.cuda.stream(stream):
        st = ctx._ba_struct(arr)
        o = ctx._ba_opts(use_huber, huber, max_iters, verbosity)
        # (collectives_at_world_one: a one-rank process group still goes through every collective -- how a one-GPU box
        # exercises the RCCL path end to end)
        use_cb = world > 1 or (collectives_at_world_one and distributed)
        cb, state = (_make_allreduce(pkg, ctx, group) if use_cb else (C.cast(None, ALLREDUCE_FN), {"error": None}))
        out = pkg.BaSummary()
        rc = ctx.L.vsl_global_bundle_adjust(ctx.h, C.byref(st), C.byref(o), cb, None, int(rank), int(world), C.byref(out))
        if state["error"] is not None:
            raise state["error"]
        ctx._ck(rc)
    return SimpleNamespace(initial_cost=out.initial_cost, final_cost=out.final_cost, iterations=out.iterations,
                           successful_steps=out.successful_steps, termination=out.termination, world=world,
                           total_ms=out.total_ms)
