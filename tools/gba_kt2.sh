#!/bin/bash
# kernel-time table + timeline of the global-BA session iteration (GPU box): bash tools/gba_kt2.sh <tag>
R=$GRAFT_REPO_ROOT; tag=${1:-gba_kt2}
mkdir -p $R/gpurun_out
python3 $R/tools/global_ba_bench.py --iters 8 > $R/gpurun_out/$tag.log 2>&1 && grep -E "marginal|vsl BA" $R/gpurun_out/$tag.log | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tools/global_ba_bench.py --iters 6 > $R/gpurun_out/$tag.prof.log 2>&1
python3 $R/tools/kt_timeline.py $R/gpurun_out/$tag 'bal_prep_kernel<false>' > $R/gpurun_out/$tag.timeline.txt
tail -1 $R/gpurun_out/$tag.timeline.txt
