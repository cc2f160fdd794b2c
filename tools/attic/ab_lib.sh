#!/bin/bash
# A/B of two library builds on ONE box: tools/attic/ab_lib.sh <variant name> <grep pattern> [rounds]
V=$1; PAT=$2; R=${3:-2}
for r in $(seq $R); do
  echo "-- current"; timeout 300 python tools/layer_times.py --bs 256 2>&1 | grep -E "$PAT"
  echo "-- $V"; SC2_LIB=tools/variants/lib_$V.so timeout 300 python tools/layer_times.py --bs 256 2>&1 | grep -E "$PAT"
done
