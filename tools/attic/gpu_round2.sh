#!/bin/bash
# Round-2 GPU session: parity tests, smoke, the driver's bench line (K=20) and the K=100 line, layer timings.
# Usage (from repo root, on the GPU box): bash tools/attic/gpu_round2.sh <tag> [steps: tests smoke bench20 bench100 layers prof]
TAG=${1:-r02a}; shift
STEPS=${@:-tests smoke bench20 bench100 layers}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8   # bench.py sets it in-process, but under rocprofv3 the runtime may initialise before Python runs
export TMPDIR=/tmp
ROOT=$(pwd)
for s in $STEPS; do
  case $s in
    tests) timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/gpu_tests.log; tail -5 $OUT/gpu_tests.log;;
    newtests) timeout 900 python -m pytest tests/test_gpu_benchpath.py -m gpu -x -q 2>&1 | tail -25 > $OUT/gpu_newtests.log; tail -8 $OUT/gpu_newtests.log;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log;;
    bench20) timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; cat $OUT/bench20.json; tail -3 $OUT/bench20.err;;
    bench100) timeout 900 python bench.py --no-cpu-baseline --no-bs1 > $OUT/bench100.json 2> $OUT/bench100.err; cat $OUT/bench100.json; tail -3 $OUT/bench100.err;;
    shapes) timeout 600 python tools/shape_times.py > $OUT/shape_times.log 2>&1; cat $OUT/shape_times.log;;
    layers) timeout 600 python tools/layer_times.py --bs 256 > $OUT/layer_times.log 2>&1; cat $OUT/layer_times.log;;
    prof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-bs1 --steps 20 --warmup 5 > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof_bench.err)
          for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done; head -30 $OUT/kernel_stats.csv;;
  esac
done
