#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs into profiles/<round>_pmc_traffic.json.

usage: pmc_summary.py <refresh dir (with pmc/, calib/, pmc_command.txt)> <out.json> <batch stereo frames per launch>

Per kernel: mean FETCH_SIZE / WRITE_SIZE per dispatch (summed over the XCD rows rocprofv3 emits for one dispatch), in
KiB as reported, and `bytes_per_launch` = corrected bytes: each counter is multiplied by the factor its calibration
kernel gives for the access width the stage uses (tools/probes/pmc_calib.hip streams 1 GiB once per width, so the
factor is known bytes / counter; MI355X_MICROARCH.md section HBM prescribes exactly this for widths other than 16 B
per lane, and gives 2.0 for 16-B reads).
"""
import csv
import json
import pathlib
import sys
from collections import defaultdict

STAGES = {  # stage -> (kernel, read width calibration, write width calibration)
    "response": ("min_eig_response_kernel", "calib_read_b8", "calib_write_b64"),
    "select": ("select_kernel", "calib_read_b32", "calib_write_b64"),
    "describe": ("describe_tile_kernel", "calib_read_b32", "calib_write_b64"),  # launches of >= 96 images (describe_fast_kernel below that)
    "describe_bin": ("describe_bin_kernel", "calib_read_b32", "calib_write_b64"),
    "describe_exact": ("exact_bits_kernel", "calib_read_b32", "calib_write_b64"),
    "match": ("hamming_mx_kernel", "calib_read_b32", "calib_write_b64", 2),  # forward + reverse pass = one match launch (round 3)
    "match_finalize": ("match_finalize_kernel", "calib_read_b32", "calib_write_b64"),
}
CALIB_BYTES = 1 << 30


def collect(root):
    acc = defaultdict(lambda: defaultdict(float))  # (kernel, counter) -> {dispatch: value}
    for f in pathlib.Path(root).rglob("*counter_collection.csv"):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0].split("<")[0].strip().split()[-1]
                acc[(name, row["Counter_Name"])][(str(f), row["Dispatch_Id"])] += float(row["Counter_Value"])
    return acc


def main():
    root, out, batch = pathlib.Path(sys.argv[1]), pathlib.Path(sys.argv[2]), int(sys.argv[3])
    cal = collect(root / "calib")
    factors = {}
    for kern in ("calib_read_b8", "calib_read_b32", "calib_read_b128"):
        d = cal.get((kern, "FETCH_SIZE"))
        if d:
            factors[kern] = CALIB_BYTES / (sum(d.values()) / len(d) * 1024)
    d = cal.get(("calib_write_b64", "WRITE_SIZE"))
    if d:
        factors["calib_write_b64"] = CALIB_BYTES / (sum(d.values()) / len(d) * 1024)
    acc = collect(root / "pmc")
    kernels = {}
    for stage, spec in STAGES.items():
        kern, rcal, wcal = spec[:3]
        per_launch = spec[3] if len(spec) > 3 else 1   # dispatches of this kernel that make up one launch of the stage
        ent = {"kernel": kern}
        if per_launch > 1:
            ent["dispatches_per_launch"] = per_launch
        tot = corrected = 0.0
        for ctr, calname in (("FETCH_SIZE", rcal), ("WRITE_SIZE", wcal)):
            d = acc.get((kern, ctr))
            if not d:
                continue
            mean = per_launch * sum(d.values()) / len(d)
            ent[f"{ctr}_KiB"] = round(mean, 1)
            ent[f"{ctr}_dispatches"] = len(d)
            ent[f"{ctr}_factor"] = round(factors.get(calname, 1.0), 4)
            tot += mean
            corrected += mean * factors.get(calname, 1.0)
        if tot:
            ent["bytes_per_launch_uncorrected"] = int(tot * 1024)
            ent["bytes_per_launch"] = int(corrected * 1024)
            kernels[stage] = ent
    cmd = (root / "pmc_command.txt").read_text().strip() if (root / "pmc_command.txt").exists() else "?"
    doc = {
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no tracing) of `python3 %s` "
                "(tools/refresh_profiles.sh): %d stereo frames = %d images per launch.  Counter values in KiB as reported; "
                "`bytes_per_launch` applies the calibration factors below (known bytes / counter for a 1 GiB single-pass "
                "stream of the same access width, tools/probes/pmc_calib.hip; MI355X_MICROARCH.md HBM section: 2.0 for "
                "16-B-per-lane reads, other widths to be calibrated in the caller's own pattern)." % (cmd, batch, 2 * batch),
        "batch_stereo_frames": batch,
        "calibration_factors": {k: round(v, 4) for k, v in factors.items()},
        "kernels": kernels,
    }
    out.write_text(json.dumps(doc, indent=1) + "\n")
    print(json.dumps(doc["calibration_factors"]), json.dumps({k: v.get("bytes_per_launch") for k, v in kernels.items()}))


if __name__ == "__main__":
    main()
