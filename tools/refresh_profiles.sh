#!/bin/bash
# Regenerates the measurements behind profiles/r01_*: run on the GPU box from the repo root
# (gpurun -- 'bash tools/refresh_profiles.sh'), then `bash tools/install_profiles.sh` here (tools/README.md).
set -e
R=$PWD
mkdir -p gpurun_out/refresh
python bench.py > gpurun_out/refresh/bench_line.json 2> gpurun_out/refresh/bench.err
cd /tmp && export TMPDIR=/tmp
# kernel durations: one stream of 128 stereo frames per launch (the isolated durations of the roofline), and the
# default two-stream run
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/refresh/kt1 -- python3 $R/bench.py --batch 512 --streams 1 --cpu-frames 0 --no-ba --no-gba --no-e2e > $R/gpurun_out/refresh/kt1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/refresh/kt2 -- python3 $R/bench.py --cpu-frames 0 --no-ba --no-gba --no-e2e > $R/gpurun_out/refresh/kt2.log 2>&1
# HBM traffic: separate counter passes, no tracing
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/refresh/pmc/fetch -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-ba --no-gba --no-e2e --profile-steps 1 --batch 512 --streams 1 > $R/gpurun_out/refresh/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/refresh/pmc/write -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-ba --no-gba --no-e2e --profile-steps 1 --batch 512 --streams 1 > $R/gpurun_out/refresh/pmc_w.log 2>&1
