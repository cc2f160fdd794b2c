#!/bin/bash
# rocprofv3 kernel stats of the stage-1 training step -> gpurun_out/<tag>_train_kernel_stats.csv ; usage: tools/train_prof.sh r03
tag=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$PWD}   # (resolved before the cd below)
out=$ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_train -- python3 $ROOT/bench.py --mode train --steps 8 --warmup 3 > $out/${tag}_train_prof_bench.json 2> $out/${tag}_train_prof.err
f=$(find $out/prof_train -name '*kernel_stats.csv' | head -1)
cp "$f" $out/${tag}_train_kernel_stats.csv
head -12 $out/${tag}_train_kernel_stats.csv | cut -c1-150
