#!/bin/bash
# A/B of an environment switch of the library on ONE box, interleaved rounds: tools/ab_env.sh VAR "<grep pattern>" [rounds]
VAR=$1; PAT=$2; R=${3:-3}
for r in $(seq $R); do
  echo "-- $VAR=0"; env $VAR=0 timeout 300 python tools/layer_times.py --bs 256 2>&1 | grep -E "$PAT"
  echo "-- $VAR=1"; env $VAR=1 timeout 300 python tools/layer_times.py --bs 256 2>&1 | grep -E "$PAT"
done
