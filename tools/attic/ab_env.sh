#!/bin/bash
# A/B of an environment switch of the library on ONE box, interleaved rounds:
#   tools/attic/ab_env.sh VAR "<values>" "<grep pattern>" [rounds]
VAR=$1; VALS=$2; PAT=$3; R=${4:-3}
for r in $(seq $R); do
  for v in $VALS; do
    echo "-- $VAR=$v"; env $VAR=$v timeout 300 python tools/layer_times.py --bs 256 2>&1 | grep -E "$PAT"
  done
done
