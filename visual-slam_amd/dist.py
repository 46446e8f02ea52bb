"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
GPU box, "gloo" in the CPU tests).

What shards:
  * per-frame path (detect / describe / match / BoW): independent streams -> each rank owns whole
    streams, NO data-path collective ("replicas only"); only the benchmark's timing uses a MAX
    all-reduce.
  * global bundle adjustment: landmarks (with their observations) are partitioned into contiguous
    ranges balanced by observation count; every rank reduces its range into a partial reduced camera
    system [S | g | cost] (vsl_ba_linearize with lm_first / lm_count) and ONE sum all-reduce per LM
    iteration produces the full system on every rank.  Local BA (S is 108 x 108) stays on one GPU.
"""
import os

import numpy as np


def env_rank_world():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init(backend):
    """Initialise torch.distributed from the torchrun environment (rendezvous on 127.0.0.1)."""
    import torch.distributed as dist
    rank, world, local_rank = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl":
            import torch
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of a python float (benchmark timing contract)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, device="cpu"):
    """Every rank's python float, in rank order (the benchmark line lists per-rank times so that a straggler shows)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(x.item()) for x in parts]


def stream_seeds(rank, n_streams_per_rank, base=10):
    """Seeds of the synthetic streams a rank owns: disjoint across ranks, stable across world sizes."""
    return [base + 1000 * rank + i for i in range(n_streams_per_rank)]


def landmark_ranges(obs_per_landmark, world):
    """Contiguous landmark ranges [first, first+count) per rank, balanced by observation count.

    Every landmark belongs to exactly one range; ranges may be empty when there are fewer landmarks
    than ranks."""
    obs = np.asarray(obs_per_landmark, np.int64)
    L = len(obs)
    csum = np.concatenate([[0], np.cumsum(obs)])
    total = int(csum[-1])
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        cuts.append(int(np.searchsorted(csum, target, side="left")))
    cuts.append(L)
    cuts = np.maximum.accumulate(np.minimum(cuts, L))
    return [(int(cuts[r]), int(cuts[r + 1] - cuts[r])) for r in range(world)]


def allreduce_sum_(tensor):
    """In-place SUM all-reduce of the packed [S | g | cost] buffer (f64)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    return tensor
