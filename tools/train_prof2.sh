#!/bin/bash
# rocprofv3 kernel stats of the STAGE-2 training step -> gpurun_out/<tag>_train2_kernel_stats.csv ; usage: tools/train_prof2.sh r05
tag=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
out=$ROOT/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_train2 -- python3 $ROOT/bench.py --mode train --stage 2 --steps 8 --warmup 3 > $out/${tag}_train2_prof_bench.json 2> $out/${tag}_train2_prof.err
f=$(find $out/prof_train2 -name '*kernel_stats.csv' | head -1)
cp "$f" $out/${tag}_train2_kernel_stats.csv
rm -rf $out/prof_train2
head -40 $out/${tag}_train2_kernel_stats.csv | cut -c1-170
