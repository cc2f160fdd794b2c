#!/bin/bash
# Diagnostic: per-kernel average durations of the pipelined bench with and without the coder chains running
# (bench.py --diag-skip-coder 1).  Usage on the GPU box: bash tools/diag_coder.sh
export GPU_MAX_HW_QUEUES=8   # bench.py sets it in-process, but under rocprofv3 the runtime may initialise before Python runs
export TMPDIR=/tmp; ROOT=$(pwd); OUT=gpurun_out/diag_coder; mkdir -p $OUT
for m in 0 1; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/m$m -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --diag-skip-coder $m > $ROOT/$OUT/m$m.json 2> $ROOT/$OUT/m$m.err)
  cp $(find $OUT/m$m -name "*kernel_stats.csv" | head -1) $OUT/stats_m$m.csv
done
