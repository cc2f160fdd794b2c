#!/bin/bash
# One GPU-box session: parity tests, smoke, bench line, rocprofv3 kernel trace of the same bench command.
# Usage (from repo root, on the GPU box): bash tools/attic/gpu_round.sh <tag> [bench args]
TAG=${1:-r01}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8   # bench.py sets it in-process, but under rocprofv3 the runtime may initialise before Python runs
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $OUT/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
python tools/layer_times.py --bs 256 > $OUT/layer_times.log 2>&1
ROOT=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o bench -- python3 $ROOT/bench.py --no-cpu-baseline "$@" > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof_bench.err)
find $OUT/prof -name "*kernel_stats*" | head -3
for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
ls -la $OUT $OUT/prof 2>/dev/null | head -30
tail -3 $OUT/gpu_tests.log; cat $OUT/smoke.log | tail -2; cat $OUT/bench.json; tail -5 $OUT/bench.err; head -25 $OUT/kernel_stats.csv
