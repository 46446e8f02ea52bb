#!/bin/bash
# A/B of matcher builds on one box: tools/match_ab.sh <lib> [<lib> ...]   (libraries under visual-slam_amd/exp/, each run twice, alternating)
for rep in 1 2; do
  for so in "$@"; do
    echo "== $so"
    VSL_SO=$so timeout -k 10 120 python tools/match_probe.py 2>&1 | grep "match " || exit 1
  done
done
