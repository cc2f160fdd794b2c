#!/bin/bash
# HBM traffic of the bottleneck forward of a --workload step (PMC, separate passes).  Usage on the GPU box: bash tools/pmc_workload.sh <tag> <workload> [...]
TAG=${1:-r04w}; shift
export GPU_MAX_HW_QUEUES=10
export TMPDIR=/tmp
ROOT=$(pwd)
for W in "$@"; do
  OUT=gpurun_out/${TAG}_$W
  mkdir -p $OUT
  for C in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/$OUT/pmc_$C -o pmc -- python3 $ROOT/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/pmc_$C.log 2>&1)
  done
  python3 tools/pmc_workload.py $OUT $W gpurun_out/${TAG}_traffic_workloads.json | tee $OUT/traffic.txt
done
