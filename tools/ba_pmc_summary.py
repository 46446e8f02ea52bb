#!/usr/bin/env python3
"""Fold the counter_collection CSVs of tools/ba_pmc.sh into profiles/<round>_ba_pmc_traffic.json: per workload (local BA at
7 / 10 keyframes, global BA at 1000 cameras) and kernel the mean FETCH_SIZE / WRITE_SIZE per dispatch (KiB as reported)
and bytes = FETCH_SIZE x 2.0 + WRITE_SIZE x 1.0; per workload the kernels of ONE LM iteration added up next to the
algorithmic bytes of SURVEY.md 8(d) for that problem (n_obs * 24 + n_lms * 24 + n_cams * 56 + 128 in; S + rhs +
n_lms * 96 out, S in the form the solver stores it)."""
import csv
import importlib
import json
import pathlib
import sys
from collections import defaultdict

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def kernel_id(full):
    s = full.strip().strip('"')
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and not s.startswith("(anonymous namespace)", i):
            cut = i
            break
    s = s[:cut].replace("(anonymous namespace)::", "").replace("void ", "").strip()
    return s.split("::")[-1].replace(" ", "")


def collect(root, prefix):
    acc = defaultdict(lambda: defaultdict(float))
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in pathlib.Path(root, "%s_%s" % (prefix, ctr)).rglob("*counter_collection.csv"):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    acc[(kernel_id(row["Kernel_Name"]), row["Counter_Name"])][(str(f), row["Dispatch_Id"])] += float(row["Counter_Value"])
    out = {}
    for (kern, ctr), d in acc.items():
        e = out.setdefault(kern, {})
        e[ctr + "_KiB"] = round(sum(d.values()) / len(d), 1)
        e[ctr + "_dispatches"] = len(d)
    for kern, e in out.items():
        e["bytes_per_dispatch"] = int((2.0 * e.get("FETCH_SIZE_KiB", 0.0) + e.get("WRITE_SIZE_KiB", 0.0)) * 1024)
    return out


def algorithmic(d, s_doubles):
    n_obs, n_lms, n_cams = len(d["obs_cam"]), len(d["points"]), len(d["poses"])
    n = 6 * int((d["cam_fixed"] == 0).sum())
    return {"n_obs": n_obs, "n_lms": n_lms, "n_cams": n_cams, "unknowns_reduced": n,
            "bytes_in": n_obs * 24 + n_lms * 24 + n_cams * 56 + 128, "bytes_out": 8 * s_doubles + 8 * n + 96 * n_lms,
            "bytes_per_iteration": n_obs * 24 + n_lms * 24 + n_cams * 56 + 128 + 8 * s_doubles + 8 * n + 96 * n_lms}


def main():
    root, out = pathlib.Path(sys.argv[1]), pathlib.Path(sys.argv[2])
    synth = importlib.import_module("__graft_entry__").load_package() and importlib.import_module("visual_slam_amd.synth")
    doc = {"note": __doc__.replace("\n", " "), "workloads": {}}
    # kernels of one LM iteration (one dispatch each per iteration unless a count is given)
    local_iter = {"baf_schur_kernel<false,1>": 1, "baf_schur_kernel<false,2>": 1, "baf_schur_kernel<false,3>": 1,
                  "baf_finish_kernel": 1, "baf_chol_kernel": 1, "baf_step_kernel": 1, "baf_decide_kernel": 1}
    for name, n_kf in (("local7", 7), ("local10", 10)):
        d = synth.ba_problem(4, n_kf=n_kf, n_lms=20000)
        n = 6 * int((d["cam_fixed"] == 0).sum())
        k = collect(root, name)
        per_iter = sum(v["bytes_per_dispatch"] * local_iter[kk] for kk, v in k.items() if kk in local_iter)
        alg = algorithmic(d, n * n)
        doc["workloads"][name] = {"command": "tools/local_ba_probe.py %d" % n_kf, "algorithmic": alg, "kernels": k,
                                  "iteration_kernels": [kk for kk in k if kk in local_iter],
                                  "bytes_per_iteration_measured": per_iter,
                                  "measured_over_algorithmic": round(per_iter / alg["bytes_per_iteration"], 2)}
    dg = synth.ba_problem(5, n_kf=500, n_lms=100000, loop_radius=200.0, max_range=15.0)
    k = collect(root, "global")
    # per iteration of the large-system path: every kernel's share = its dispatches / the dispatches of the kernel that
    # runs once per iteration (recompute form: bal_prep_kernel<false>; stored-blocks chain: ba_linearize_kernel)
    unit = "bal_prep_kernel<false>" if "bal_prep_kernel<false>" in k else "ba_linearize_kernel"
    lin = k.get(unit, {}).get("FETCH_SIZE_dispatches", 0)
    per_iter = 0.0
    shares = {}
    for kk, v in k.items():
        if lin and not kk.startswith("ba_pair_") and not kk.startswith("__amd"):
            calls = v.get("FETCH_SIZE_dispatches", v.get("WRITE_SIZE_dispatches", 0)) / lin
            shares[kk] = round(calls, 2)
            per_iter += calls * v["bytes_per_dispatch"]
    band = None
    try:
        log = (root / "global_FETCH_SIZE.log").read_text()
        for ln in log.splitlines():
            if "bandwidth" in ln and band is None:
                band = ln.strip()
    except OSError:
        pass
    n = 6 * int((dg["cam_fixed"] == 0).sum())
    doc["workloads"]["global"] = {"command": "tools/global_ba_bench.py --iters 6   (the session path bench.py times)",
                                  "per_iteration_unit": unit,
                                  "algorithmic_dense_S": algorithmic(dg, n * n), "band_note": band, "kernels": k,
                                  "dispatches_per_linearisation": shares, "bytes_per_iteration_measured": int(per_iter)}
    out.write_text(json.dumps(doc, indent=1) + "\n")
    for w, e in doc["workloads"].items():
        print(w, "measured bytes per iteration", e["bytes_per_iteration_measured"], e.get("measured_over_algorithmic"))
        for kk, v in sorted(e["kernels"].items(), key=lambda kv: -kv[1]["bytes_per_dispatch"])[:8]:
            print("   %-40s %12d B per dispatch" % (kk, v["bytes_per_dispatch"]))


if __name__ == "__main__":
    main()
