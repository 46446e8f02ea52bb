#!/bin/bash
# HBM traffic of the bundle-adjustment kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (no tracing)
# of tools/local_ba_probe.py (BASELINE configs[2], 7 and 10 keyframes: the fused iteration of ba_fused.hip) and of
# tools/global_ba_bench.py (configs[4] through the session path: the kernels of ba_large.h / ba.hip / chol.hip), corrected like
# profiles/rNN_pmc_traffic.json (FETCH_SIZE x 2.0, WRITE_SIZE x 1.0: tools/probes/pmc_calib.hip).
# On the GPU box: bash tools/ba_pmc.sh r04   ->  gpurun_out/r04_ba_pmc_traffic.json
R=$GRAFT_REPO_ROOT; TAG=${1:-r04}
O=$R/gpurun_out/${TAG}_ba_pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/local7_$ctr -- python3 $R/tools/local_ba_probe.py 7 > $O/local7_$ctr.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/local10_$ctr -- python3 $R/tools/local_ba_probe.py 10 > $O/local10_$ctr.log 2>&1 || exit 1
  timeout -k 10 600 rocprofv3 --pmc $ctr --output-format csv -d $O/global_$ctr -- python3 $R/tools/global_ba_bench.py --iters 6 > $O/global_$ctr.log 2>&1 || exit 1
done
python3 $R/tools/ba_pmc_summary.py $O $R/gpurun_out/${TAG}_ba_pmc_traffic.json
