#!/bin/bash
# per-kernel durations of the matcher launch for library builds (run on the GPU box): tools/match_kt.sh <lib> [<lib> ...]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  name=$(basename $so .so)
  export VSL_SO=$R/$so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mkt_$name -- python3 $R/tools/match_probe.py > $R/gpurun_out/mkt_$name.log 2>&1 || exit 1
  echo "== $name"
  python3 - $R/gpurun_out/mkt_$name <<'PY'
import csv,sys,glob,os
f=max(glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    if "hamming" in r["Name"] or "match_" in r["Name"]:
        print("%-40s calls %5s avg %9.1f us min %9.1f"%(r["Name"].replace("void ","")[:40], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
