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
