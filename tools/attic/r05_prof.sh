#!/bin/bash
# rocprofv3 kernel stats of one bench.py invocation: bash tools/attic/r05_prof.sh <tag> <name> <bench args...>
TAG=$1; NAME=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
ROOT=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_$NAME -o p -- python3 $ROOT/bench.py "$@" --no-cpu-baseline > $ROOT/$OUT/prof_$NAME.json 2> $ROOT/$OUT/prof_$NAME.err)
F=$(find $OUT/prof_$NAME -name "*kernel_stats.csv" | head -1)
cp $F $OUT/${NAME}_kernel_stats.csv 2>/dev/null
head -25 $OUT/${NAME}_kernel_stats.csv | cut -c1-200
rm -rf $OUT/prof_$NAME
