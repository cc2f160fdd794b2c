#!/bin/bash
# Stage-1 Entropic-Student training step: bench line + rocprofv3 kernel stats of the same command.
TAG=${1:-r02_train}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8; export TMPDIR=/tmp; ROOT=$(pwd)
timeout 900 python bench.py --mode train --steps 10 --warmup 3 > $OUT/train_bench.json 2> $OUT/train_bench.err; cat $OUT/train_bench.json; tail -3 $OUT/train_bench.err
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o train -- python3 $ROOT/bench.py --mode train --steps 6 --warmup 2 > $ROOT/$OUT/prof_train.json 2> $ROOT/$OUT/prof_train.err)
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/train_kernel_stats.csv; done; head -25 $OUT/train_kernel_stats.csv | cut -c1-220
