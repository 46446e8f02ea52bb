set -e
R=$PWD; O=$R/gpurun_out/refresh; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LITE="--cpu-frames 0 --no-ba --no-gba --no-e2e --stream-seconds 0 --gen-workers 1"
python3 $R/bench.py --streams 1 --batch 512 --passes 1 --steps 1 --warmup 0 --cpu-frames 0 --no-ba --no-gba --no-e2e --stream-seconds 0 > /dev/null 2>&1
PMCARGS="--streams 1 --batch 512 --passes 1 --steps 3 --warmup 1 --profile-steps 1 $LITE"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d $O/fsq -- python3 $R/bench.py $PMCARGS > $O/fsq.log 2>&1
echo fsq done
