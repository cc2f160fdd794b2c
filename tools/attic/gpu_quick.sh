#!/bin/bash
# Quick GPU iteration: selected tests + head / layer timings + the two bench lines.
# Usage: bash tools/attic/gpu_quick.sh <tag> "<pytest -k expression or empty>" [steps: test head layers bench20 bench100]
TAG=${1:-q}; KEXPR=${2:-}; shift; shift
STEPS=${@:-test head}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
for s in $STEPS; do
  case $s in
    test) timeout 1200 python -m pytest tests -m gpu -x -q -k "$KEXPR" 2>&1 | tail -15 > $OUT/test.log; tail -6 $OUT/test.log;;
    alltests) timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/gpu_tests.log; tail -5 $OUT/gpu_tests.log;;
    head) timeout 600 python tools/head_times.py > $OUT/head_times.log 2>&1; cat $OUT/head_times.log;;
    layers) timeout 600 python tools/layer_times.py --bs 256 > $OUT/layer_times.log 2>&1; cat $OUT/layer_times.log;;
    bench20) timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-bs1 > $OUT/bench20.json 2> $OUT/bench20.err; python tools/bench_brief.py $OUT/bench20.json; tail -3 $OUT/bench20.err;;
    bench100) timeout 900 python bench.py --no-cpu-baseline --no-bs1 > $OUT/bench100.json 2> $OUT/bench100.err; python tools/bench_brief.py $OUT/bench100.json; tail -3 $OUT/bench100.err;;
  esac
done
