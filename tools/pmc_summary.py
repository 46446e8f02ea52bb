#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (one pass per counter) into profiles/<round>_pmc_traffic.json.

usage: pmc_summary.py <dir with *_counter_collection.csv (searched recursively)> <out.json> [batch]

Per kernel name: mean counter value per dispatch (summed over the XCD/instance rows rocprofv3 emits for one
dispatch).  Values are the raw counter units (KiB for FETCH_SIZE / WRITE_SIZE).
"""
import csv
import json
import pathlib
import sys
from collections import defaultdict

STAGES = {
    "response": "min_eig_response_kernel",
    "select": "select_kernel",
    "describe": "describe_fast_kernel",
    "describe_exact": "exact_bits_kernel",
    "match": "hamming_mfma_kernel",
    "match_finalize": "match_finalize_kernel",
}


def main():
    root, out = pathlib.Path(sys.argv[1]), pathlib.Path(sys.argv[2])
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 128
    # (kernel, counter) -> {dispatch id: value}
    acc = defaultdict(lambda: defaultdict(float))
    for f in root.rglob("*counter_collection.csv"):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].split("(")[0].split("<")[0].strip()
                name = name.split()[-1]
                acc[(name, row["Counter_Name"])][(str(f), row["Dispatch_Id"])] += float(row["Counter_Value"])
    kernels = {}
    for stage, kern in STAGES.items():
        ent = {"kernel": kern}
        tot = 0.0
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = acc.get((kern, ctr))
            if not d:
                continue
            mean = sum(d.values()) / len(d)
            ent[f"{ctr}_KiB"] = round(mean, 1)
            ent[f"{ctr}_dispatches"] = len(d)
            tot += mean
        if tot:
            ent["bytes_per_launch_uncorrected"] = int(tot * 1024)
            kernels[stage] = ent
    doc = {
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, no tracing) of `python3 bench.py "
                "--steps 3 --warmup 1 --cpu-frames 0 --no-ba --no-gba --no-e2e --profile-steps 1 --batch 128 --streams 1` (tools/refresh_profiles.sh; batch 128 stereo frames = 256 images "
                "per launch). Raw counter values in KiB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the "
                "bytes of a 16 B/lane stream and is uncalibrated for other widths -- the kernels here load 1-4 B per "
                "lane, so `bytes_per_launch_uncorrected` = (FETCH + WRITE) * 1024 is a lower bound.",
        "batch_stereo_frames": batch,
        "kernels": kernels,
    }
    out.write_text(json.dumps(doc, indent=1) + "\n")
    print(json.dumps(kernels, indent=1))


if __name__ == "__main__":
    main()
