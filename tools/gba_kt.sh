#!/bin/bash
# kernel-time table of the global-BA bench under rocprofv3 (run on the GPU box): bash tools/gba_kt.sh <tag> [bench args]
R=$GRAFT_REPO_ROOT; tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tools/global_ba_bench.py --iters 6 --single-call "$@" > $R/gpurun_out/$tag.log 2>&1
grep -E "marginal|vsl BA" $R/gpurun_out/$tag.log | tail -2
python3 - $R/gpurun_out/$tag <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:8]:
    print("%-58s calls %5s avg %9.1f us  total %8.2f ms"%(r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
