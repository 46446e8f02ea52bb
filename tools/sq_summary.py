#!/usr/bin/env python3
"""SQ counters of the matcher kernels (tools/refresh_profiles.sh, passes mm1..mm3) -> profiles/<round>_matcher_sq_counters.json:
per kernel variant the mean counter value per dispatch (millions), summed over the rows rocprofv3 emits per dispatch."""
import csv
import json
import pathlib
import sys
from collections import defaultdict


def main():
    root, out = pathlib.Path(sys.argv[1]), pathlib.Path(sys.argv[2])
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))
    for tag in ("mm1", "mm2", "mm3"):
        for f in (root / tag).rglob("*counter_collection.csv"):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = row["Kernel_Name"]
                    if "hamming_m" not in k:
                        continue
                    if "hamming_mx" in k:  # round 3: a match launch is a forward and a reverse dispatch of this kernel
                        name = "fp4_block_scaled_reverse_pass" if ("<true>" in k or "ILb1" in k) else "fp4_block_scaled_forward_pass"
                    elif "true" in k or "Lb1" in k:
                        name = "int8_staggered"
                    else:
                        name = "int8_lockstep"
                    acc[name][row["Counter_Name"]][(str(f), row["Dispatch_Id"])] += float(row["Counter_Value"])
    doc = {"note": "rocprofv3 --pmc (no tracing) of tools/match_probe.py: 512 stereo pairs of 1500 keypoints per launch; mean per "
                   "DISPATCH in millions (the FP4 matcher is a forward dispatch over all 1500 queries and a reverse dispatch "
                   "over the ~300 listed columns per pair; the int8 variants run both full directions in one dispatch).  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, "
                   "SQ_VALU_MFMA_BUSY_CYCLES and SQ_VALU_MFMA_COEXEC_CYCLES count cycles (MI355X_MICROARCH.md).",
           "kernels": {}}
    for name, ctrs in acc.items():
        doc["kernels"][name] = {c: round(sum(d.values()) / len(d) / 1e6, 3) for c, d in sorted(ctrs.items())}
        doc["kernels"][name]["dispatches_averaged"] = max(len(d) for d in ctrs.values())
    out.write_text(json.dumps(doc, indent=1) + "\n")
    print(json.dumps(doc["kernels"], indent=1))


if __name__ == "__main__":
    main()
