#!/bin/bash
# A/B of the global-BA session iteration on the GPU box: recompute form (ba_large.h) against the stored-blocks chain
# (VSL_BA_NO_FUSED=1), then a kernel-time table of the recompute form.   bash tools/gba_ab.sh <tag>
R=$GRAFT_REPO_ROOT; tag=${1:-gba_ab}
mkdir -p $R/gpurun_out
python3 $R/tools/global_ba_bench.py --iters 8 > $R/gpurun_out/$tag.new.log 2>&1 && \
VSL_BA_NO_FUSED=1 python3 $R/tools/global_ba_bench.py --iters 8 > $R/gpurun_out/$tag.old.log 2>&1 && \
grep -H -E "marginal|vsl BA|cost" $R/gpurun_out/$tag.new.log $R/gpurun_out/$tag.old.log | tail -12
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tools/global_ba_bench.py --iters 6 > $R/gpurun_out/$tag.prof.log 2>&1
python3 - $R/gpurun_out/$tag <<'PY'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:14]:
    print("%-58s calls %5s avg %9.1f us  total %8.2f ms"%(r['Name'][:58], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
