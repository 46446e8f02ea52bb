"""The committed bench line of the newest round (profiles/rNN_bench_line.json = the stdout of `python bench.py` on an
MI355X) carries every key the benchmark contract names, with consistent values; and bench.py's command line still
parses the driver's flags."""
import json
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
ROUND = sorted(p.name.split("_")[0] for p in (ROOT / "profiles").glob("r[0-9][0-9]_bench_line.json"))[-1]


def test_committed_bench_line_has_the_contract_keys():
    d = json.loads((ROOT / "profiles" / (ROUND + "_bench_line.json")).read_text())
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "frames/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None  # BASELINE.md holds no published number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = frames of all ranks / time of the K timed steps
    frames = d["n_gpus"] * d["config"]["stereo_frames_per_step_per_gpu"] * d["steps"]
    assert abs(frames / (d["ms_per_step"] * 1e-3 * d["steps"]) - d["value"]) < 1e-3 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-4
    # achieved = algorithmic bytes per launch / average launch duration
    assert abs(r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9 - r["achieved"]) < 1e-2 * r["achieved"]
    # K1 must at least read every image byte of the launch (its own outputs are a few MB; the 56 B per keypoint of the
    # stage's algorithmic figure are written by the kernels behind it)
    images = 2 * d["config"]["stereo_frames_per_launch"]
    assert r["traffic"] is None or 0.95 * images * 752 * 480 <= r["traffic"] <= 3 * r["algorithmic_bytes_per_launch"]
    # the sustained-rate and upload-inclusive figures of round 2
    assert d["config"]["inputs_resident_in_hbm"] is True and d["config"]["h2d_in_timed_region"] is False
    assert d["config"]["timed_region_s"] >= 1.0
    assert 0 < d["value_incl_upload"] < d["value"] and d["streaming"]["outputs_equal_resident_run"] is True
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["unit"] == d["unit"]


def test_pmc_traffic_profile_matches_the_bench_launch_size():
    # bench.py fills roofline.traffic from this file only when it was collected at the launch size it runs at
    d = json.loads((ROOT / "profiles" / (ROUND + "_bench_line.json")).read_text())
    pm = json.loads((ROOT / "profiles" / (ROUND + "_pmc_traffic.json")).read_text())
    assert pm["batch_stereo_frames"] == d["config"]["stereo_frames_per_launch"]
    k = pm["kernels"][d["roofline"]["kernel"]]
    assert abs(k["bytes_per_launch_uncorrected"] - (k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024) < 2048
    # the calibrated figure (FETCH_SIZE x its measured factor + WRITE_SIZE x its factor) is what the bench line carries
    want = (k["FETCH_SIZE_KiB"] * k["FETCH_SIZE_factor"] + k["WRITE_SIZE_KiB"] * k["WRITE_SIZE_factor"]) * 1024
    assert abs(k["bytes_per_launch"] - want) < 1e-3 * want


def test_sq_counter_summaries_hold_physically_possible_values():
    # a summary that folded two kernels into one row carried 53.6 resident waves per SIMD (the
    # hardware holds 8) and 0.55 SIMD-cycles per vector instruction (a SIMD issues at most one per cycle).  Every
    # committed summary from round 4 on is checked; the tool asserts the same when it writes one.
    for p in sorted((ROOT / "profiles").glob("r*_sq_counters.json")):
        if int(p.name[1:3]) < 4:
            continue
        doc = json.loads(p.read_text())
        for name, k in doc["kernels"].items():
            assert k.get("mean_resident_waves_per_simd", 0) <= 8.4, (p.name, name, k)
            assert k.get("simd_cycles_per_valu_instruction", 9) >= 1.0, (p.name, name, k)
    sys.path.insert(0, str(ROOT / "tools"))
    import frame_sq_summary as fs
    # whole kernel names: no substring hits across kernels, the two matcher instances apart
    assert fs.row_name("(anonymous namespace)::match_select_kernel(int const*, int const*)") == "match_select"
    assert fs.row_name("void select_kernel<false>(unsigned long const*, int*)") == "select"
    assert fs.row_name("void hamming_mx_kernel<true>(unsigned long const*)") == "match_reverse"
    assert fs.row_name("void hamming_mx_kernel<false>(unsigned long const*)") == "match_forward"
    assert fs.row_name("exact_bits_kernel(unsigned char const*)") is None


def test_bench_command_line_accepts_the_driver_flags():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup"):
        assert flag in out.stdout


def _run_bench(args, env_extra, timeout=300):
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra)
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_gpus_n_without_a_launcher_starts_n_ranks():
    # `python bench.py --gpus 2` (no torchrun around it) must produce 2 ranks -- it used to run one
    # rank and print "n_gpus": 1.  Rehearsed on the CPU: gloo, a sleeping step (--rehearse-plumbing), the real launcher path.
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse-plumbing"], {"VSL_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout          # ONE line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["rehearsal"] is True and d["value"] is None
    assert len(d["per_rank_ms_per_step"]) == 2
    assert d["per_rank_ms_per_step"][1] > d["per_rank_ms_per_step"][0]          # the planted straggler shows
    assert d["ms_per_step"] >= max(d["per_rank_ms_per_step"]) - 1e-3            # MAX over ranks


def test_gpus_n_that_disagrees_with_the_launcher_fails():
    r = _run_bench(["--gpus", "8", "--rehearse-plumbing"], {"WORLD_SIZE": "1", "RANK": "0", "VSL_BENCH_BACKEND": "gloo"}, timeout=120)
    assert r.returncode != 0 and "does not match WORLD_SIZE" in (r.stdout + r.stderr)
