#!/bin/bash
# Round-4 GPU sessions.  Usage (repo root, on the GPU box): bash tools/attic/gpu_r4.sh <tag> <step> [<step> ...]
#   steps: w2tests | ab:<variant>[:<k_times args>] | tests | smoke | bench20 | bench100 | ktimes | layers | head | prof | pmc | kt:<k_times args>
TAG=${1:-r04a}; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
export TMPDIR=/tmp
ROOT=$(pwd)
for s in "$@"; do
  case $s in
    w2tests) timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv2x2_win or conv3x3_win or conv1x1_win or gdn512 or conv2x2_gdn" 2>&1 | tail -15 | tee $OUT/w2tests.log;;
    ktests) timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q 2>&1 | tail -15 | tee $OUT/ktests.log;;
    ab:*) IFS=: read -r _ V ARGS <<< "$s"
          for r in 1 2; do
            echo "-- current" | tee -a $OUT/ab_$V.txt; timeout 300 python tools/k_times.py $ARGS 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_$V.txt
            echo "-- $V" | tee -a $OUT/ab_$V.txt; SC2_LIB=tools/variants/lib_$V.so timeout 300 python tools/k_times.py $ARGS 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab_$V.txt
          done;;
    kt:*) ARGS=${s#kt:}; timeout 300 python tools/k_times.py $ARGS 2>&1 | grep -v amdgpu.ids | tee $OUT/k_times.txt;;
    ktimes) timeout 300 python tools/k_times.py --head 2>&1 | grep -v amdgpu.ids | tee $OUT/k_times.txt;;
    tests) timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/gpu_tests.log; tail -4 $OUT/gpu_tests.log;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log;;
    bench20) timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; python tools/bench_brief.py $OUT/bench20.json; tail -3 $OUT/bench20.err;;
    bench100) timeout 900 python bench.py --no-cpu-baseline --no-bs1 > $OUT/bench100.json 2> $OUT/bench100.err; python tools/bench_brief.py $OUT/bench100.json;;
    layers) timeout 600 python tools/layer_times.py --bs 256 > $OUT/layer_times.txt 2>&1; tail -36 $OUT/layer_times.txt;;
    head) timeout 600 python tools/head_times.py > $OUT/head_times.txt 2>&1; tail -50 $OUT/head_times.txt;;
    prof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-bs1 --steps 20 --warmup 5 > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof_bench.err)
          for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/bench_kernel_stats.csv; done; rm -rf $OUT/prof; head -12 $OUT/bench_kernel_stats.csv | cut -c1-160;;
    pmc) bash tools/pmc_round.sh ${TAG}_pmc > $OUT/pmc_round.log 2>&1; tail -20 $OUT/pmc_round.log; bash tools/pmc_mfma.sh ${TAG}_pmc_mfma > $OUT/pmc_mfma.log 2>&1; tail -30 $OUT/pmc_mfma.log;;
    *) echo "unknown step $s";;
  esac
done
