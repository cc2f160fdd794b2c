#!/bin/bash
# A/B of several library builds on ONE box: tools/attic/ab_multi.sh "<grep pattern>" <rounds> <variant names...>  ("cur" = the in-tree build)
PAT=$1; R=$2; shift; shift
for r in $(seq $R); do
  for V in "$@"; do
    if [ "$V" == "cur" ]; then L=""; else L="tools/variants/lib_$V.so"; fi
    echo "-- $V"; SC2_LIB=$L timeout 300 python tools/layer_times.py --bs 256 2>&1 | grep -E "$PAT"
  done
done
