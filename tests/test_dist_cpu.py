"""CPU, world_size 2, gloo: the N > 1 plumbing (visual-slam_amd/dist.py).

* per-frame path: ranks own disjoint streams, timing is a MAX all-reduce, no data-path collective;
* global BA: landmark ranges partition the problem, and summing the per-rank partial reduced camera
  systems with ONE all-reduce reproduces the single-rank system.  The per-rank partials are computed
  by the CPU oracle here (no GPU in this container); the GPU version of the same additivity property
  is tests/test_ba_gpu.py::test_linearize_partition_is_additive."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    import importlib
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    entry.load_package()
    vdist = importlib.import_module("visual_slam_amd.dist")
    synth = importlib.import_module("visual_slam_amd.synth")
    orc = entry.load_oracle()
    r, w, _ = vdist.init("gloo")
    assert (r, w) == (rank, world)
    # timing contract: MAX over ranks
    t = vdist.max_over_ranks(1.0 + rank)
    assert t == float(world)
    # streams: disjoint seeds
    seeds = vdist.stream_seeds(rank, 3)
    gathered = [None] * world
    dist.all_gather_object(gathered, seeds)
    flat = [s for g in gathered for s in g]
    assert len(set(flat)) == len(flat) == 3 * world
    # global BA: partial [S | g | cost] per landmark range, one SUM all-reduce
    d = synth.ba_problem(7, n_kf=6, n_lms=1200, loop_radius=5.0)
    arr = orc.BaArrays(d["poses"], d["cam_fixed"], d["cam_intr"], d["intr"], d["points"], d["obs_cam"], d["obs_lm"],
                       d["obs_uv"], d["cam_model"])
    counts = np.bincount(arr.obs_lm, minlength=len(arr.points))
    ranges = vdist.landmark_ranges(counts, world)
    first, cnt = ranges[rank]
    S, g, c = orc.ba_linearize(arr, lm_first=first, lm_count=cnt)
    n = len(g)
    buf = torch.from_numpy(np.concatenate([S.ravel(), g, [c]]))
    vdist.allreduce_sum_(buf)
    vdist.barrier()
    if rank == 0:
        Sf, gf, cf = orc.ba_linearize(arr)
        out = buf.numpy()
        ok = (np.allclose(out[:n * n].reshape(n, n), Sf, rtol=0, atol=1e-9 * np.abs(Sf).max())
              and np.allclose(out[n * n:n * n + n], gf, rtol=0, atol=1e-9 * np.abs(gf).max())
              and abs(out[-1] - cf) <= 1e-12 * cf)
        (Path(out_dir) / "ok").write_text("1" if ok else "0")
    dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok").read_text() == "1"


def test_landmark_ranges_partition(vsl):
    import importlib
    vdist = importlib.import_module("visual_slam_amd.dist")
    rng = np.random.default_rng(0)
    obs = rng.integers(2, 30, 1000)
    for world in (1, 2, 3, 8):
        rs = vdist.landmark_ranges(obs, world)
        assert len(rs) == world and rs[0][0] == 0 and sum(c for _, c in rs) == 1000
        for (a, ca), (b, _) in zip(rs[:-1], rs[1:]):
            assert a + ca == b
        loads = [obs[a:a + c].sum() for a, c in rs]
        assert max(loads) - min(loads) <= 2 * obs.max()
    # fewer landmarks than ranks: empty ranges allowed, still a partition
    rs = vdist.landmark_ranges([5, 5], 8)
    assert sum(c for _, c in rs) == 2 and all(c >= 0 for _, c in rs)
