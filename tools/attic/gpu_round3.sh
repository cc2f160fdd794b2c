#!/bin/bash
# Round-3 GPU session: parity tests, smoke, the driver's bench line (K=20) and the K=100 line, layer / head timings, rocprofv3
# kernel stats, PMC passes.  Usage (from repo root, on the GPU box): bash tools/attic/gpu_round3.sh <tag> [steps...]
TAG=${1:-r03f}; shift
STEPS=${@:-tests smoke bench20 bench100 layers head prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
ROOT=$(pwd)
for s in $STEPS; do
  case $s in
    tests) timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/gpu_tests.log; tail -3 $OUT/gpu_tests.log;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log;;
    bench20) timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; python tools/bench_brief.py $OUT/bench20.json; tail -3 $OUT/bench20.err;;
    bench100) timeout 900 python bench.py --no-cpu-baseline --no-bs1 > $OUT/bench100.json 2> $OUT/bench100.err; python tools/bench_brief.py $OUT/bench100.json;;
    layers) timeout 600 python tools/layer_times.py --bs 256 > $OUT/layer_times.txt 2>&1; tail -30 $OUT/layer_times.txt;;
    head) timeout 600 python tools/head_times.py > $OUT/head_times.txt 2>&1; tail -50 $OUT/head_times.txt;;
    prof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-bs1 --steps 20 --warmup 5 > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof_bench.err)
          for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/bench_kernel_stats.csv; done; rm -rf $OUT/prof; head -12 $OUT/bench_kernel_stats.csv | cut -c1-160;;
    pmc) bash tools/pmc_round.sh ${TAG}_pmc > $OUT/pmc_round.log 2>&1; tail -20 $OUT/pmc_round.log; bash tools/pmc_mfma.sh ${TAG}_pmc_mfma > $OUT/pmc_mfma.log 2>&1; tail -20 $OUT/pmc_mfma.log;;
  esac
done
