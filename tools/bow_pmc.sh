#!/bin/bash
# HBM traffic of the BoW kernels (K8 / K9) on the 1.1 M-node vocabulary: FETCH_SIZE and WRITE_SIZE in separate rocprofv3
# --pmc passes of tools/bow_probe.py (M = 10,000 candidates from host arrays), corrected like profiles/rNN_pmc_traffic.json
# (FETCH_SIZE x 2.0, WRITE_SIZE x 1.0: tools/probes/pmc_calib.hip).  On the GPU box: bash tools/bow_pmc.sh r03
R=$GRAFT_REPO_ROOT; TAG=${1:-r03}
O=$R/gpurun_out/${TAG}_bow_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/bow_probe.py --no-oracle --M 100 > /dev/null 2>&1   # writes the vocabulary file outside the profiler
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/bow_probe.py --no-oracle --M 10000 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/tools/bow_probe.py --no-oracle --M 10000 > $O/write.log 2>&1
python3 - $O $R/gpurun_out/${TAG}_bow_pmc_traffic.json <<'PY'
import csv, glob, json, sys
from collections import defaultdict
root, out = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float))
for f in glob.glob(root + '/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0]
        if name.startswith('bow_'):
            acc[(name, r['Counter_Name'])][(f, r['Dispatch_Id'])] += float(r['Counter_Value'])
doc = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/bow_probe.py --M 10000 on the k = 10, L = 6 "
               "vocabulary; KiB as reported, bytes = FETCH_SIZE x 2.0 + WRITE_SIZE x 1.0 (the calibration of "
               "profiles/rNN_pmc_traffic.json).  bow_score_lds_kernel: the LAST dispatches are the M = 10,000 launches "
               "(algorithmic bytes 178.5 MB = 12 B per candidate word + the query + 8 B per score)", "kernels": {}}
for kern in sorted(set(k for k, _ in acc)):
    ent = {}
    for ctr, fac in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
        d = acc.get((kern, ctr))
        if d:
            vals = [v for _, v in sorted(d.items(), key=lambda kv: (kv[0][0], int(kv[0][1])))]
            ent[ctr + "_KiB_mean"] = round(sum(vals) / len(vals), 1)
            ent[ctr + "_KiB_max"] = round(max(vals), 1)
            ent[ctr + "_dispatches"] = len(vals)
            ent.setdefault("bytes_largest_dispatch", 0)
            ent["bytes_largest_dispatch"] += int(max(vals) * 1024 * fac)
    doc["kernels"][kern] = ent
json.dump(doc, open(out, 'w'), indent=1)
print(json.dumps(doc["kernels"], indent=1))
PY
