#!/bin/bash
# On the GPU box: which rendered laps close their loop with the reference's defaults (no test hooks)?
#   bash tools/loop_sweep.sh  ->  gpurun_out/loop_sweep.log   (one line per configuration: GPU build, device-resident)
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/loop_sweep.log
: > $out
run() {  # room ppm look frames step radius [extra flags]
  local room=$1 ppm=$2 look=$3 n=$4 step=$5 rad=$6; shift 6
  local line
  line=$(ROOM=$room PPM=$ppm LOOK=$look timeout -k 10 400 bash $R/tools/e2e_probe.sh $n $step $rad --fused "$@" 2>&1 | tail -1)
  echo "room $room ppm $ppm look $look n $n step $step radius $rad $* :: $(echo "$line" | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('kf %d ate %.3f lost %d reloc %d loops %d gba %d fps %.0f' % (d['keyframes'], d['ate_rmse_m'], d['tracking_lost'], d['relocalized'], d['loops_closed'], d['global_ba_runs'], d['frames_per_s']))
except Exception as e:
    print('ERR', e)
")" | tee -a $out
}
"$@"
