"""GPU: the multi-GPU global-BA path (vsl_ba_session + visual-slam_amd/ba_dist.py).

* world size 1: the session-based LM loop reproduces vsl_bundle_adjust and the oracle;
* world size 2 (two processes sharing the one GPU of the test box, gloo backend -- NCCL refuses two
  ranks on one device): landmark-range partition + SUM all-reduce of the packed reduced camera system
  gives the same trajectory (iterations, termination) and the same optimum as the single-rank solve.
  On the 8-GPU node the same code runs with backend "nccl" (RCCL over xGMI) and the buffers stay in HBM."""
import importlib
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _arr(orc, d):
    return orc.BaArrays(d["poses"], d["cam_fixed"], d["cam_intr"], d["intr"], d["points"], d["obs_cam"], d["obs_lm"],
                        d["obs_uv"], d["cam_model"])


def _problem(synth):
    return synth.ba_problem(81, n_kf=30, n_lms=4000, loop_radius=5.0)


def test_session_world1_matches_single_call_and_oracle(vsl, orc, synth):
    import torch
    ba_dist = importlib.import_module("visual_slam_amd.ba_dist")
    d = _problem(synth)
    ctx = vsl.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    a_sess, a_one, a_cpu = _arr(orc, d), _arr(orc, d), _arr(orc, d)
    s = ba_dist.bundle_adjust_distributed(vsl, ctx, a_sess, max_iters=8)
    # (the single call takes large maps through the session solver itself since round 4; "ba_no_fused" keeps it on the
    # operator-by-operator chain over stored blocks: the independent path this test compares with)
    ctx.set_diagnostic("ba_no_fused", 1)
    try:
        s1 = ctx.bundle_adjust(a_one, max_iters=8)
    finally:
        ctx.set_diagnostic("ba_no_fused", 0)
    s2 = orc.bundle_adjust(a_cpu, max_iters=8)
    assert (s.iterations, s.termination, s.successful_steps) == (s1.iterations, s1.termination, s1.successful_steps)
    assert (s.iterations, s.termination) == (s2.iterations, s2.termination)
    assert s.initial_cost == pytest.approx(s1.initial_cost, rel=1e-12)
    assert s.final_cost == pytest.approx(s1.final_cost, rel=1e-9)
    assert s.final_cost == pytest.approx(s2.final_cost, rel=1e-7)
    # 58 free cameras -> the large-system Schur path.  Its sums are fixed-order gathers (bit-reproducible since the
    # round-2 rewrite), but the session path reduces per landmark RANGE and adds the partial systems, i.e. in another
    # order than the single call: ill-conditioned depths may move by more than the well-conditioned bulk
    assert np.allclose(a_sess.poses, a_one.poses, rtol=0, atol=1e-7)
    dp = np.abs(a_sess.points - a_one.points).max(1)
    assert (dp < 1e-6).mean() > 0.97 and dp.max() < 0.05
    ctx.close()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as entry
    vsl = entry.load_package()
    orc = entry.load_oracle()
    synth = importlib.import_module("visual_slam_amd.synth")
    ba_dist = importlib.import_module("visual_slam_amd.ba_dist")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = _problem(synth)
    arr = _arr(orc, d)
    ctx = vsl.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    s = ba_dist.bundle_adjust_distributed(vsl, ctx, arr, max_iters=8)
    np.savez(Path(out_dir) / ("rank%d.npz" % rank), poses=arr.poses, points=arr.points,
             meta=np.array([s.iterations, s.termination, s.successful_steps, s.world]), cost=np.array([s.initial_cost, s.final_cost]))
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()


def test_session_world2_gloo_shared_gpu(tmp_path, vsl, orc, synth, ctx):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    # every rank ends with the same full solution
    assert np.array_equal(r0["poses"], r1["poses"]) and np.array_equal(r0["points"], r1["points"])
    assert r0["meta"].tolist() == r1["meta"].tolist() and r0["meta"][3] == 2
    d = _problem(synth)
    one = _arr(orc, d)
    ctx.set_diagnostic("ba_no_fused", 1)   # the stored-blocks single call: an independent path (see above)
    try:
        s1 = ctx.bundle_adjust(one, max_iters=8)
    finally:
        ctx.set_diagnostic("ba_no_fused", 0)
    assert r0["meta"][:3].tolist() == [s1.iterations, s1.termination, s1.successful_steps]
    assert r0["cost"][0] == pytest.approx(s1.initial_cost, rel=1e-12)
    assert r0["cost"][1] == pytest.approx(s1.final_cost, rel=1e-8)
    assert np.allclose(r0["poses"], one.poses, rtol=0, atol=1e-7)
    dp = np.abs(r0["points"] - one.points).max(1)
    assert (dp < 1e-6).mean() > 0.97 and dp.max() < 0.05


def test_rccl_backend_collectives_at_world_size_one():
    # the "nccl" (= RCCL) code path of dist.py / bench.py / ba_dist.py: init with device_id, barrier, MAX and SUM
    # all-reduce of f64 device buffers.  World size 1 is what a one-GPU box allows; N > 1 runs on the driver's node.
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "rccl_smoke.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl smoke ok" in r.stdout and "rccl session ok" in r.stdout, r.stdout + r.stderr


def _write_problem(path, d):
    import struct
    with open(path, "wb") as f:
        f.write(struct.pack("iii", len(d["poses"]), len(d["points"]), len(d["obs_cam"])))
        for a, t in ((d["poses"], np.float64), (d["cam_fixed"], np.uint8), (d["intr"], np.float64),
                     (d["points"], np.float64), (d["obs_cam"], np.int32), (d["obs_lm"], np.int32), (d["obs_uv"], np.float64)):
            f.write(np.ascontiguousarray(a, t).tobytes())


def test_cpp_global_bundle_adjustment_through_rccl(tmp_path, orc, synth, ctx):
    # the C++ caller: visnav::global_bundle_adjustment -> vsl_global_bundle_adjust with ncclAllReduce on the solver's
    # stream (include/visnav_amd/rccl_world.h); a one-rank RCCL communicator is what a one-GPU box allows
    import subprocess
    exe = tmp_path / "global_ba_rccl_test"
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O2", "-I", str(ROOT / "include"), str(ROOT / "tests/cpp/global_ba_rccl_test.cpp"),
           "-o", str(exe), "-L", str(ROOT / "visual-slam_amd"), "-lvslam_hip", "-Wl,-rpath," + str(ROOT / "visual-slam_amd"),
           "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = _problem(synth)
    _write_problem(tmp_path / "ba.bin", d)
    env = dict(os.environ, VISNAV_AMD_FORCE_RCCL="1", VISNAV_AMD_NCCL_ID_FILE=str(tmp_path / "nccl_id"))
    r = subprocess.run([str(exe), str(tmp_path / "ba.bin"), str(tmp_path / "out.bin"), "8"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0 and "rccl on, rank 0 of 1" in r.stdout, r.stdout + r.stderr
    out = np.fromfile(tmp_path / "out.bin", np.float64)
    nc, nl = len(d["poses"]), len(d["points"])
    poses, points = out[:7 * nc].reshape(nc, 7), out[7 * nc:].reshape(nl, 3)
    one = _arr(orc, d)
    s1 = ctx.bundle_adjust(one, max_iters=8)
    assert s1.iterations == 8 or s1.termination != 0
    assert np.allclose(poses, one.poses, rtol=0, atol=1e-7)
    dp = np.abs(points - one.points).max(1)
    assert (dp < 1e-6).mean() > 0.97 and dp.max() < 0.05


def test_linearize_at_baseline_config_4_matches_oracle(vsl, orc, synth):
    # BASELINE.json configs[4] at FULL size, against the ORACLE: 500 keyframes = 1000 cameras,
    # ~97 k landmarks, ~880 k observations; one ba_linearize -- the reduced camera system S (5988 x 5988), the gradient g
    # and the robustified cost -- GPU (the large-system gather path) vs orc.ba_linearize (map_utils.h / loop_closure_utils.h
    # functor + the restated Schur elimination), 1e-9 relative to the largest entry; and through two landmark ranges
    # (the multi-GPU partition): the partial systems add up to the same S.
    d = synth.ba_problem(5, n_kf=500, n_lms=100000, loop_radius=200.0, max_range=15.0)
    arr = _arr(orc, d)
    c = vsl.Context(0)
    S, g, cost = c.ba_linearize(arr)
    eS, eg, ecost = orc.ba_linearize(arr)
    n = 6 * int((d["cam_fixed"] == 0).sum())
    assert S.shape == eS.shape == (n, n) and n == 5988
    assert cost == pytest.approx(ecost, rel=1e-12)
    scale = np.abs(eS).max()
    assert np.abs(S - eS).max() <= 1e-9 * scale
    assert np.abs(g - eg).max() <= 1e-9 * np.abs(eg).max()
    assert np.abs(S - S.T).max() <= 1e-9 * scale
    L = len(arr.points)
    obs_per_lm = np.bincount(arr.obs_lm, minlength=L)
    ranges = vsl_dist(vsl).landmark_ranges(obs_per_lm, 2)
    Ss, gs, cs = 0, 0, 0
    for first, count in ranges:
        S1, g1, c1 = c.ba_linearize(arr, lm_first=first, lm_count=count)
        Ss, gs, cs = Ss + S1, gs + g1, cs + c1
    assert np.abs(Ss - eS).max() <= 1e-9 * scale and cs == pytest.approx(ecost, rel=1e-12)
    c.close()


def vsl_dist(vsl):
    return importlib.import_module("visual_slam_amd.dist")


def test_global_ba_full_size_properties(vsl, orc, synth):
    # BASELINE configs[4] scale (500 keyframes = 1000 cameras, ~1e5 landmarks, ~9e5 observations), size-independent
    # properties instead of the oracle (which needs minutes): band form == dense form (same LM trajectory: iterations,
    # termination, costs to 1e-9), large cost reduction, fixed cameras untouched, single call == session path.  The linear
    # solve itself is checked at this size (5988 unknowns, half bandwidth 221: ||S x - b|| / ||b|| <= 1e-10) in
    # tests/test_chol_gpu.py::test_band_solve_at_global_ba_size
    import torch
    ba_dist = importlib.import_module("visual_slam_amd.ba_dist")
    d = synth.ba_problem(5, n_kf=500, n_lms=100000, loop_radius=200.0, max_range=15.0)
    c = vsl.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    a_band, a_dense = _arr(orc, d), _arr(orc, d)
    s_band = ba_dist.bundle_adjust_distributed(vsl, c, a_band, max_iters=6)
    c.set_diagnostic("ba_force_dense", 1)
    try:
        s_dense = ba_dist.bundle_adjust_distributed(vsl, c, a_dense, max_iters=6)
    finally:
        c.set_diagnostic("ba_force_dense", 0)
    assert (s_band.iterations, s_band.termination, s_band.successful_steps) == (s_dense.iterations, s_dense.termination, s_dense.successful_steps)
    assert s_band.initial_cost == pytest.approx(s_dense.initial_cost, rel=1e-12)
    assert s_band.final_cost == pytest.approx(s_dense.final_cost, rel=1e-9)
    assert s_band.final_cost < 0.3 * s_band.initial_cost
    # the default form of this closed loop is the CYCLIC band (cameras in trajectory order, distances around the ring:
    # less than half the bandwidth of the best linear order); the linear band form ("ba_no_cyclic": reverse
    # Cuthill-McKee) is the same optimisation
    a_cyc = _arr(orc, d)
    ba_dist.bundle_adjust_distributed(vsl, c, a_cyc, max_iters=1)
    _, banded_cyc, bw_cyc = c.last_ba_layout()
    c.set_diagnostic("ba_no_cyclic", 1)
    try:
        a_lin = _arr(orc, d)
        s_lin = ba_dist.bundle_adjust_distributed(vsl, c, a_lin, max_iters=6)
        _, banded_lin, bw_lin = c.last_ba_layout()
    finally:
        c.set_diagnostic("ba_no_cyclic", 0)
    assert banded_cyc == 2 and banded_lin == 1 and 2 * bw_cyc < bw_lin
    assert (s_lin.iterations, s_lin.termination, s_lin.successful_steps) == (s_band.iterations, s_band.termination, s_band.successful_steps)
    assert s_lin.final_cost == pytest.approx(s_band.final_cost, rel=1e-9)
    assert np.allclose(a_lin.poses, a_band.poses, rtol=0, atol=1e-7)
    fixed = d["cam_fixed"].astype(bool)
    assert np.array_equal(a_band.poses[fixed], d["poses"][fixed])
    assert np.allclose(a_band.poses, a_dense.poses, rtol=0, atol=1e-6)
    # observations in ARBITRARY order (the set-up then sorts them by landmark and builds its indices on several host
    # threads): the same optimisation
    perm = np.random.default_rng(3).permutation(len(d["obs_cam"]))
    d_sh = dict(d)
    for k in ("obs_cam", "obs_lm", "obs_uv"):
        d_sh[k] = np.ascontiguousarray(d[k][perm])
    a_sh = _arr(orc, d_sh)
    s_sh = ba_dist.bundle_adjust_distributed(vsl, c, a_sh, max_iters=6)
    assert (s_sh.iterations, s_sh.termination, s_sh.successful_steps) == (s_band.iterations, s_band.termination, s_band.successful_steps)
    assert s_sh.final_cost == pytest.approx(s_band.final_cost, rel=1e-9)
    assert np.allclose(a_sh.poses, a_band.poses, rtol=0, atol=1e-7)
    # single call (the stored-blocks chain under "ba_no_fused"; by default it IS the session path) == session path
    a_one = _arr(orc, d)
    c.set_diagnostic("ba_no_fused", 1)
    try:
        s_one = c.bundle_adjust(a_one, max_iters=6)
    finally:
        c.set_diagnostic("ba_no_fused", 0)
    a_def = _arr(orc, d)
    s_def = c.bundle_adjust(a_def, max_iters=6)
    assert (s_def.iterations, s_def.termination, s_def.final_cost) == (s_band.iterations, s_band.termination, s_band.final_cost)
    assert (s_one.iterations, s_one.termination) == (s_band.iterations, s_band.termination)
    assert s_one.final_cost == pytest.approx(s_band.final_cost, rel=1e-9)
    c.close()


def test_bench_gpus_2_runs_the_global_ba_all_reduce_leg():
    # `python bench.py --gpus N` must carry BASELINE configs[4] THROUGH THE COLLECTIVE at N > 1 -- the
    # global-BA leg used to be gated to world == 1, so the one exchange step of the path (the J^T J all-reduce) was in no
    # command the driver runs.  Two ranks (gloo: host-memory all-reduce; RCCL needs one GPU per rank) sharing the one GPU
    # of the box, a reduced problem (60 keyframes) and a short frame loop: n_gpus 2 AND global_ba.world 2, the final cost
    # equal to the world-1 solve of the same problem to 1e-9, the same LM trajectory.
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["VSL_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "32",
                        "--scenes", "8", "--passes", "1", "--cpu-frames", "0", "--stream-seconds", "0", "--no-ba", "--no-e2e",
                        "--no-bow", "--gen-workers", "2", "--gba-kf", "60", "--gba-lms", "6000", "--profile-steps", "1"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and len(d["per_rank"]["frames_per_s"]) == 2 and d["value"] > 0
    g = d["global_ba"]
    assert g["world"] == 2 and len(g["per_rank_ms_per_iteration"]) == 2
    assert g["allreduce_bytes_per_iteration"] > 8 * 6 * 118     # at least the right-hand side of the 118 free cameras
    assert g["iterations"] == g["iterations_world1"]
    assert g["final_cost_rel_diff_vs_world1"] <= 1e-9
    assert g["max_pose_diff_vs_world1"] <= 1e-7
    assert g["ms_per_lm_iteration_marginal"] > 0


def _session_solve(vsl, synth, d, no_fused, max_iters=8):
    import torch
    ba_dist = importlib.import_module("visual_slam_amd.ba_dist")
    ctx = vsl.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    if no_fused:
        ctx.set_diagnostic("ba_no_fused", 1)
    a = vsl.BaArrays.from_dict(d)
    s = ba_dist.bundle_adjust_distributed(vsl, ctx, a, max_iters=max_iters)
    ctx.close()
    return s, a


def test_recompute_form_matches_the_stored_blocks_form(vsl, synth):
    """The session iteration that evaluates observations on the fly (ba_large.h) against the chain over stored
    r / F / E blocks ("ba_no_fused"): the same LM trajectory, the same optimum; and a solve is reproducible bit for bit."""
    d = _problem(synth)
    s_new, a_new = _session_solve(vsl, synth, d, False)
    s_old, a_old = _session_solve(vsl, synth, d, True)
    assert (s_new.iterations, s_new.termination, s_new.successful_steps) == (s_old.iterations, s_old.termination, s_old.successful_steps)
    assert s_new.initial_cost == pytest.approx(s_old.initial_cost, rel=1e-12)
    assert s_new.final_cost == pytest.approx(s_old.final_cost, rel=1e-9)
    assert np.allclose(a_new.poses, a_old.poses, rtol=0, atol=1e-7)
    dp = np.abs(a_new.points - a_old.points).max(1)
    assert (dp < 1e-6).mean() > 0.97 and dp.max() < 0.05
    s_again, a_again = _session_solve(vsl, synth, d, False)
    assert s_again.final_cost == s_new.final_cost
    assert np.array_equal(a_again.poses, a_new.poses) and np.array_equal(a_again.points, a_new.points)


def test_a_landmark_with_more_observations_than_a_workgroup_takes_the_stored_blocks_form(vsl, orc, synth):
    """The recompute-form kernels give a landmark's observations one thread each of a 512-thread workgroup; a landmark
    observed more often than that (here: its observations repeated) must fall back to the stored-blocks chain, not fail."""
    d = dict(_problem(synth))
    lm = int(np.bincount(d["obs_lm"]).argmax())
    idx = np.nonzero(d["obs_lm"] == lm)[0]
    reps = 520 // len(idx) + 1
    extra = np.tile(idx, reps)
    for k in ("obs_cam", "obs_lm", "obs_uv"):
        d[k] = np.ascontiguousarray(np.concatenate([d[k], d[k][extra]]))
    assert (d["obs_lm"] == lm).sum() > 512
    s_def, a_def = _session_solve(vsl, synth, d, False, max_iters=5)
    s_old, a_old = _session_solve(vsl, synth, d, True, max_iters=5)
    # the same kernels ran both times (equal up to the summation order of pair lists longer than PAIR_SORT_MAX, which
    # the repeated observations produce here: those keep their fill order, ba.hip ba_pair_sort_kernel)
    assert s_def.final_cost == pytest.approx(s_old.final_cost, rel=1e-12)
    assert np.allclose(a_def.poses, a_old.poses, rtol=0, atol=1e-11)
    a_cpu = _arr(orc, d)
    s_cpu = orc.bundle_adjust(a_cpu, max_iters=5)
    assert (s_def.iterations, s_def.termination) == (s_cpu.iterations, s_cpu.termination)
    assert s_def.final_cost == pytest.approx(s_cpu.final_cost, rel=1e-7)
