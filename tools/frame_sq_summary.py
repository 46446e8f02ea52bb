#!/usr/bin/env python3
"""SQ counters of the per-frame kernels (tools/refresh_profiles.sh, pass `fsq`) -> profiles/<round>_frame_sq_counters.json:
per kernel the mean counter value per dispatch and what they say about the binding resource.  Quad-cycle counters
(SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_*) are summed over all waves.  A SIMD's vector ALU executes one wave's
instruction at a time, so  launch duration in cycles * 1024 SIMDs / SQ_INSTS_VALU  = SIMD-cycles available per VALU
instruction: compare it with the measured issue cost of the kernel's instruction mix (tools/probes/valu_rate.hip: 2.3
cycles for fp32 / integer ops, 4.2 for fp64, conversions and DPP moves, 8.2 for v_sqrt_f32) -- a kernel whose figure
equals its mix average is VALU-issue-bound, and the HBM roofline fraction of such a kernel says little.  Durations come
from the kernel trace of the one-stream run (kt1), the clock is taken as the 2.4 GHz peak (a lower sustained clock
means fewer cycles were available)."""
import csv
import json
import pathlib
import sys
from collections import defaultdict

import re

# WHOLE kernel names (namespace / return type / parameter list stripped, template arguments kept): substring matching
# used to fold match_select_kernel into "select" and both hamming_mx_kernel instances into one row
NAMES = {"min_eig_response_kernel": "response", "select_kernel": "select", "describe_tile_kernel": "describe",
         "hamming_mx_kernel<false>": "match_forward", "hamming_mx_kernel<true>": "match_reverse",
         "match_select_kernel": "match_select"}
MAX_WAVES_PER_SIMD = 8   # gfx950: a figure above it means two kernels were folded into one row


def kernel_id(full):
    """'void (anonymous namespace)::hamming_mx_kernel<true>(unsigned long const*, ...)' -> 'hamming_mx_kernel<true>'."""
    s = full.strip().strip('"')
    depth, cut = 0, len(s)
    for i, ch in enumerate(s):          # the parameter list starts at the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and not s.startswith("(anonymous namespace)", i):
            cut = i
            break
    s = s[:cut].replace("(anonymous namespace)::", "")
    s = re.sub(r"^(void|int)\s+", "", s.strip())
    return s.split("::")[-1].replace(" ", "")


def row_name(full):
    kid = kernel_id(full)     # with template arguments first (the two matcher instances), then the bare name
    return NAMES.get(kid) or NAMES.get(kid.split("<")[0])


def main():
    root, out = pathlib.Path(sys.argv[1]), pathlib.Path(sys.argv[2])
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for f in (root / "fsq").rglob("*counter_collection.csv"):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row_name(row["Kernel_Name"])
                if name:
                    acc[name][row["Counter_Name"]][(str(f), row["Dispatch_Id"])] += float(row["Counter_Value"])
    dur = {}
    for f in (root / "kt1").rglob("*kernel_stats.csv"):
        for row in csv.DictReader(open(f, newline="")):
            name = row_name(row["Name"])
            if name:
                dur[name] = float(row["AverageNs"])
    doc = {"note": __doc__.split("->")[1].strip(), "clock_ghz_assumed": 2.4, "kernels": {}}
    for name, ctrs in acc.items():
        k = {c: round(sum(d.values()) / len(d) / 1e6, 3) for c, d in sorted(ctrs.items())}
        k["dispatches_averaged"] = max(len(d) for d in ctrs.values())
        if name in dur:
            cyc = dur[name] * 2.4 * 1024 / 1e6   # SIMD-cycles of the launch, in millions
            k["avg_launch_us"] = round(dur[name] / 1e3, 1)
            if "SQ_INSTS_VALU" in k:
                k["simd_cycles_per_valu_instruction"] = round(cyc / k["SQ_INSTS_VALU"], 2)
            if "SQ_WAVE_CYCLES" in k:
                k["mean_resident_waves_per_simd"] = round(4 * k["SQ_WAVE_CYCLES"] / cyc, 2)
                assert k["mean_resident_waves_per_simd"] <= MAX_WAVES_PER_SIMD * 1.05, (name, k)
        if "SQ_WAVE_CYCLES" in k and k["SQ_WAVE_CYCLES"] > 0:
            for c, label in (("SQ_WAIT_INST_ANY", "wave_time_waiting_share"), ("SQ_ACTIVE_INST_VALU", "wave_time_valu_share"),
                             ("SQ_ACTIVE_INST_LDS", "wave_time_lds_share")):
                if c in k:
                    k[label] = round(k[c] / k["SQ_WAVE_CYCLES"], 3)
        doc["kernels"][name] = k
    out.write_text(json.dumps(doc, indent=1) + "\n")
    print(json.dumps(doc["kernels"], indent=1))


if __name__ == "__main__":
    main()
