#!/bin/bash
# HBM traffic of the bottleneck kernels from PMC counters (separate passes; --kernel-trace only, no other trace domains).
# Usage on the GPU box: bash tools/pmc_round.sh <tag>
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=10   # bench.py sets it in-process, but under rocprofv3 the runtime may initialise before Python runs
export TMPDIR=/tmp
ROOT=$(pwd)
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/$OUT/pmc_$C -o pmc -- python3 $ROOT/tools/layer_times.py --bs 256 --iters 4 > $ROOT/$OUT/pmc_$C.log 2>&1)
  find $OUT/pmc_$C -name "*counter_collection.csv" | head -2
done
python3 tools/pmc_parse.py $OUT > $OUT/traffic.txt 2>&1
cat $OUT/traffic.txt | tail -30
