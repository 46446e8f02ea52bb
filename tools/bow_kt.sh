#!/bin/bash
# kernel-time table of the BoW probe under rocprofv3 (run on the GPU box): bash tools/bow_kt.sh <tag>
R=$GRAFT_REPO_ROOT; tag=${1:-bowkt}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tools/bow_probe.py --no-oracle > $R/gpurun_out/$tag.log 2>&1
tail -8 $R/gpurun_out/$tag.log
python3 - $R/gpurun_out/$tag <<'PY'
import csv,sys,glob,os
f=max(glob.glob(sys.argv[1]+'/*/*kernel_stats.csv'), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:14]:
    print("%-60s calls %5s avg %9.1f us min %9.1f"%(r["Name"].replace("void ","").replace("(anonymous namespace)::","")[:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
