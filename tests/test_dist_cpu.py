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


# ---- host-side rules of the C++ multi-GPU path (include/visnav_amd/device_select.h, file_rendezvous.h) -------------

def _build_rendezvous_test(tmp_path):
    import subprocess
    exe = tmp_path / "rendezvous_test"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-pthread", "-I", str(ROOT / "include"),
                        str(ROOT / "tests/cpp/rendezvous_test.cpp"), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_solver_context_and_communicator_pick_the_same_device(tmp_path):
    # RcclWorld bound the communicator to LOCAL_RANK while amd::ctx() defaulted to device 0.  Both
    # now call device_index_for(): VISNAV_AMD_DEVICE, else LOCAL_RANK, else VISNAV_AMD_RANK / RANK, modulo the count.
    import subprocess
    exe = _build_rendezvous_test(tmp_path)
    base = {k: v for k, v in os.environ.items() if k not in ("VISNAV_AMD_DEVICE", "LOCAL_RANK", "RANK", "VISNAV_AMD_RANK")}

    def pick(n, **env):
        r = subprocess.run([str(exe), "device", str(n)], capture_output=True, text=True, env=dict(base, **env))
        assert r.returncode == 0
        return int(r.stdout)

    assert pick(8) == 0
    assert pick(8, LOCAL_RANK="1") == 1                       # the torchrun-style launch of the advisor's finding
    assert pick(8, LOCAL_RANK="3", RANK="11") == 3
    assert pick(8, RANK="5") == 5
    assert pick(8, VISNAV_AMD_DEVICE="6", LOCAL_RANK="1") == 6
    assert pick(1, LOCAL_RANK="3") == 0                       # ranks rehearsing on a one-GPU box
    assert pick(4, VISNAV_AMD_RANK="6") == 2
    # and both users of the rule are wired to it (no second copy of the rule in either header)
    kp = (ROOT / "include/visnav_amd/keypoints.h").read_text()
    rw = (ROOT / "include/visnav_amd/rccl_world.h").read_text()
    assert "device_index_for(vsl_device_count())" in kp and 'getenv("VISNAV_AMD_DEVICE")' not in kp
    assert "device_index_for(ndev)" in rw and "LOCAL_RANK\"" not in rw


def test_file_rendezvous_ignores_stale_files(tmp_path):
    # a rendezvous file of an earlier (crashed / one-rank) run must never be taken for this run's.
    # Plant a well-formed stale id file and stale hello files, start ranks 1 and 2 BEFORE rank 0, and expect this
    # run's payload everywhere and a clean directory afterwards.
    import struct
    import subprocess
    import time
    exe = _build_rendezvous_test(tmp_path)
    path = tmp_path / "nccl_id"
    world = 3
    stale = struct.pack("<QQ", 0x76736c5f6e63636c, world) + struct.pack("<3Q", 0, 111, 222) + b"STALE".ljust(128, b"\0")
    path.write_bytes(stale)
    (tmp_path / "nccl_id.hello.1").write_bytes(struct.pack("<Q", 111))
    (tmp_path / "nccl_id.hello.2").write_bytes(struct.pack("<Q", 222))
    procs = {}
    for r in (1, 2):
        procs[r] = subprocess.Popen([str(exe), "meet", str(path), str(r), str(world), "x"], stdout=subprocess.PIPE,
                                    stderr=subprocess.PIPE, text=True)
    time.sleep(0.4)   # the late rank 0: ranks 1, 2 have been staring at the stale file for a while
    assert all(p.poll() is None for p in procs.values()), "a rank accepted the stale file"
    procs[0] = subprocess.Popen([str(exe), "meet", str(path), "0", str(world), "fresh-id"], stdout=subprocess.PIPE,
                                stderr=subprocess.PIPE, text=True)
    for r, p in procs.items():
        out, err = p.communicate(timeout=60)
        assert p.returncode == 0, err
        assert out.strip() == "rank %d got fresh-id" % r
    assert sorted(f.name for f in tmp_path.iterdir() if f.name.startswith("nccl_id")) == []
