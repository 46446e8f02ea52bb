"""RCCL smoke test at world size 1 (the only size a one-GPU box allows): the collectives bench.py and ba_dist.py
issue -- barrier, MAX all-reduce of the timing scalar, SUM all-reduce of a packed f64 buffer -- through the
"nccl" backend exactly as visual-slam_amd/dist.py sets it up for N > 1."""
import os
import sys
from pathlib import Path

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("LOCAL_RANK", "0")
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dist.barrier()
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 1.25
n = 600
buf = torch.arange(n * n + 3 * n + 2, dtype=torch.float64, device="cuda")
ref = buf.clone()
dist.all_reduce(buf, op=dist.ReduceOp.SUM)
assert torch.equal(buf, ref)
dist.barrier()
dist.destroy_process_group()
print("rccl smoke ok: backend nccl, world 1, f64 buffer of %d doubles" % buf.numel())

# ---- the global-BA session through the RCCL callback of ba_dist.py at world size 1: every collective of the LM loop is
# an in-place ncclAllReduce on the library's own buffer (a tensor aliasing it), on the context's stream, no host sync.
import importlib  # noqa: E402
import numpy as np  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

vsl = entry.load_package()
synth = importlib.import_module("visual_slam_amd.synth")
ba_dist = importlib.import_module("visual_slam_amd.ba_dist")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
d = synth.ba_problem(21, n_kf=30, n_lms=4000, loop_radius=5.0)
stream = torch.cuda.Stream()
ctx = vsl.Context(0, stream=stream.cuda_stream)
a_ref, a_cb = vsl.BaArrays.from_dict(d), vsl.BaArrays.from_dict(d)
s_ref = ba_dist.bundle_adjust_distributed(vsl, ctx, a_ref, max_iters=8)
s_cb = ba_dist.bundle_adjust_distributed(vsl, ctx, a_cb, max_iters=8, collectives_at_world_one=True)
assert (s_cb.iterations, s_cb.termination) == (s_ref.iterations, s_ref.termination)
assert s_cb.final_cost == s_ref.final_cost and np.array_equal(a_cb.poses, a_ref.poses) and np.array_equal(a_cb.points, a_ref.points)
ctx.close()
dist.destroy_process_group()
print("rccl session ok: %d LM iterations through in-place all-reduces, identical to the callback-free solve" % s_cb.iterations)
