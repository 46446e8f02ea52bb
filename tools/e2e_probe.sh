#!/bin/bash
# On the GPU box: render a lap, write a k=10 L=6 vocabulary, run slam_headless with the reference's default-on branches.
#   bash tools/e2e_probe.sh <n_frames> <step_m> <radius> [extra slam_headless flags]
R=$GRAFT_REPO_ROOT; N=${1:-640}; STEP=${2:-0.03}; RAD=${3:-2.674}; shift 3
LOOK=${LOOK:-0}
ROOM=${ROOM:-4.0,2.5,4.0}
PPM=${PPM:-110.0}
D=/tmp/vsl_lap_${N}_${STEP}_${RAD}_${LOOK}_${ROOM}_${PPM}
if [ ! -f $D/calib.json ]; then
python3 - <<PY
import sys, importlib, os; sys.path.insert(0, "$R"); import __graft_entry__ as e; e.load_package()
sq = importlib.import_module('visual_slam_amd.synth_sequence')
sq.render_sequence("$D", n_frames=$N, seed=1, step_m=$STEP, radius=$RAD, workers=min(16, os.cpu_count()), look_deg=$LOOK, room_half=($ROOM), px_per_m=$PPM)
PY
fi
V=/tmp/vsl_voc_k10L6_s7.txt
if [ ! -f $V ]; then
python3 - <<PY
import sys, importlib; sys.path.insert(0, "$R"); import __graft_entry__ as e; e.load_package()
s = importlib.import_module('visual_slam_amd.synth')
s.write_vocabulary_text("$V", 10, 6, *s.vocabulary_arrays(7, 10, 6))
PY
fi
$R/visual-slam_amd/slam_headless --dataset-path $D --cam-calib $D/calib.json --voc-path $V --relocalization --loop-closure "$@"
