#!/bin/bash
# Regenerates the measurements behind profiles/<round>_*: run ON the GPU box from the repo root
#   gpurun --timeout 1190 -- 'bash tools/refresh_profiles.sh r03'
# then `bash tools/install_profiles.sh r03` here (tools/README.md).  Counters are collected in their own passes,
# never together with tracing (MI355X_MICROARCH.md, rocprofv3 PMC slots).  The HBM-traffic passes come first and their
# summary is written into the box's profiles/ before the plain bench run, so the committed bench line carries the
# roofline.traffic of the SAME build.
set -e
R=$PWD
ROUND=${1:-r04}
O=$R/gpurun_out/refresh
rm -rf $O && mkdir -p $O  # (clear the LOCAL gpurun_out/refresh too before a new run: gpurun merges, it does not delete)
cd /tmp && export TMPDIR=/tmp
LITE="--cpu-frames 0 --no-ba --no-gba --no-e2e --no-bow --stream-seconds 0 --gen-workers 1"
# (no forked generator workers under the profiler: the frames are generated once here and cached under /tmp)
python3 $R/bench.py --streams 1 --batch 512 --passes 1 --steps 1 --warmup 0 --cpu-frames 0 --no-ba --no-gba --no-e2e --no-bow --stream-seconds 0 > /dev/null 2>&1
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes, plus the calibration of both counters on known byte counts
PMCARGS="--streams 1 --batch 512 --passes 1 --steps 3 --warmup 1 --profile-steps 1 $LITE"
echo "bench.py $PMCARGS" > $O/pmc_command.txt
echo pmc; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc/fetch -- python3 $R/bench.py $PMCARGS > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc/write -- python3 $R/bench.py $PMCARGS > $O/pmc_w.log 2>&1
echo calib; rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib/fetch -- $R/tools/probes/pmc_calib > $O/calib_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/calib/write -- $R/tools/probes/pmc_calib > $O/calib_w.log 2>&1
python3 $R/tools/pmc_summary.py $O $R/profiles/${ROUND}_pmc_traffic.json 512 > $O/pmc_summary.log 2>&1
# single-stream device accounting of the end-to-end leg (kernel launches and kernel time per frame); its summary goes
# into the box's profiles/ as well, so that the bench line's device_accounting block is this build's
echo e2e; (cd $R && LOOK=90 bash tools/e2e_profile.sh $ROUND > $O/e2e_profile.log 2>&1; cp gpurun_out/${ROUND}_e2e_kernel_stats.json gpurun_out/${ROUND}_e2e_kernel_stats.csv profiles/ 2>/dev/null; cp gpurun_out/${ROUND}_e2e_kernel_stats.json gpurun_out/${ROUND}_e2e_kernel_stats.csv $O/)
(cd $R && LOOK=90 LEG=stages bash tools/e2e_profile.sh $ROUND > $O/e2e_stages_profile.log 2>&1; cp gpurun_out/${ROUND}_stages_e2e_kernel_stats.json gpurun_out/${ROUND}_stages_e2e_kernel_stats.csv $O/ 2>/dev/null)
# bundle adjustment: HBM traffic of the fused local iteration and of the session kernels of the 1000-camera map (separate
# --pmc passes); into the box's profiles/ as well, so that the bench line's local_ba / global_ba rooflines carry THIS build's
echo ba_pmc; (cd $R && bash tools/ba_pmc.sh $ROUND > $O/ba_pmc.log 2>&1; cp gpurun_out/${ROUND}_ba_pmc_traffic.json $O/ 2>/dev/null; cp gpurun_out/${ROUND}_ba_pmc_traffic.json profiles/ 2>/dev/null)
cd /tmp
echo bench; (cd $R && python3 bench.py > $O/bench_line.json 2> $O/bench.err)
# BoW (K8 / K9) and the ORB front end: kernel durations on the k = 10 / L = 6 vocabulary
echo bow; rocprofv3 --kernel-trace --stats --output-format csv -d $O/bow -- python3 $R/tools/bow_probe.py --no-oracle > $O/bow.log 2>&1
# kernel durations: one stream (the isolated durations the roofline uses) and the default two-stream run
# (as long as the default run -- 18 passes x 20 steps after 5 warm-up steps -- so that the trace sees the sustained clocks the bench line's stage times see)
echo kt1; rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt1 -- python3 $R/bench.py --streams 1 --batch 512 --passes 18 --steps 20 --warmup 5 $LITE > $O/kt1.log 2>&1
echo kt2; rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt2 -- python3 $R/bench.py --passes 8 --steps 6 --warmup 2 $LITE > $O/kt2.log 2>&1
# per-frame kernels: SQ counters of the bench launch (what binds K1 / describe / select / match)
echo fsq; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d $O/fsq -- python3 $R/bench.py $PMCARGS > $O/fsq.log 2>&1
# matcher: SQ counters of the FP4 kernel and of the int8 kernel (staggered and not)
echo matcher; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU --output-format csv -d $O/mm1 -- python3 $R/tools/match_probe.py match_use_i8=1 > $O/mm1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU --output-format csv -d $O/mm2 -- python3 $R/tools/match_probe_nostagger.py > $O/mm2.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $O/mm3 -- python3 $R/tools/match_probe.py match_use_i8=1 > $O/mm3.log 2>&1
# bundle adjustment: kernel durations of the local window and of the 1000-camera map
echo ba; rocprofv3 --kernel-trace --stats --output-format csv -d $O/lba -- python3 $R/tools/local_ba_probe.py 7 > $O/lba.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gba -- python3 $R/tools/global_ba_bench.py --iters 8 > $O/gba.log 2>&1
echo refresh done
