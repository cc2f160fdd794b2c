#!/bin/bash
# Round 6 baseline pass: per-kernel timings, head timings, the two bench lines.
TAG=${1:-r06a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python tools/k_times.py > $OUT/k_times.txt 2>&1
timeout 300 python tools/head_times.py > $OUT/head_times.txt 2>&1
timeout 600 python bench.py --steps 20 > $OUT/bench20.json 2> $OUT/bench20.err
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cat $OUT/k_times.txt | tail -40
tail -50 $OUT/head_times.txt
python tools/bench_brief.py $OUT/bench20.json | head -30
python tools/bench_brief.py $OUT/bench_default.json | head -30
