#!/bin/bash
# A/B of library builds on one box with tools/stage_probe.py (per-stage times of the 512-frame launch + output checksum):
#   tools/stage_ab.sh <lib> [<lib> ...]      (libraries under visual-slam_amd/exp/, each run twice, alternating)
for rep in 1 2; do
  for so in "$@"; do
    echo "== $so"
    VSL_SO=$so timeout -k 10 200 python tools/stage_probe.py 2>&1 | grep -v "amdgpu.ids" | tail -3 || exit 1
  done
done
