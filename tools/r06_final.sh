#!/bin/bash
# Round-6 evidence session (repo root, on the GPU box): bash tools/r06_final.sh [tag] [steps...]
#   steps (default: all): tests smoke bench20 bench100 ktimes head clock prof workloads train pmc
TAG=${1:-r06f}; shift
STEPS=${@:-tests smoke bench20 bench100 ktimes head clock prof workloads train pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=10 TMPDIR=/tmp
ROOT=$(pwd)
t0=$(date +%s)
brief() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[2], round(d['value'], 1), d['unit'], round(d['ms_per_step'], 3), 'ms/step')
except Exception as e:
    print(sys.argv[2], 'failed', e)
PY
}
for s in $STEPS; do
  echo "== $(( $(date +%s) - t0 )) s: $s"
  case $s in
    tests) timeout 2700 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $OUT/gpu_tests.log; tail -4 $OUT/gpu_tests.log;;
    smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log;;
    bench20) timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err; python tools/bench_brief.py $OUT/bench20.json; tail -3 $OUT/bench20.err;;
    bench100) timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python tools/bench_brief.py $OUT/bench_default.json; tail -3 $OUT/bench_default.err;;
    ktimes) timeout 300 python tools/k_times.py --head 2>&1 | grep -v amdgpu.ids | tee $OUT/k_times.txt;;
    head) timeout 600 python tools/head_times.py > $OUT/head_times.txt 2>&1; tail -45 $OUT/head_times.txt;;
    clock) timeout 900 python tools/clock_probe.py --f32 --head 2>&1 | grep -v amdgpu.ids | tee $OUT/clock_probe.txt;;
    prof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --no-bs1 --no-secondary --steps 20 --warmup 5 > $ROOT/$OUT/prof_bench.json 2> $ROOT/$OUT/prof_bench.err)
          for f in $(find $OUT/prof -name "*kernel_stats.csv"); do cp $f $OUT/bench_kernel_stats.csv; done; rm -rf $OUT/prof; head -12 $OUT/bench_kernel_stats.csv | cut -c1-160;;
    workloads) for W in mshp224 seg513 det800x1216 fp_input; do
          timeout 600 python bench.py --workload $W --steps 40 --no-cpu-baseline > $OUT/bench_$W.json 2> $OUT/bench_$W.err; brief $OUT/bench_$W.json $W; done;;
    train) timeout 600 python bench.py --mode train --steps 20 --warmup 5 > $OUT/train_stage1.json 2> $OUT/train_stage1.err; brief $OUT/train_stage1.json train_stage1
           timeout 600 python bench.py --mode train --stage 2 --steps 20 --warmup 5 > $OUT/train_stage2.json 2> $OUT/train_stage2.err; brief $OUT/train_stage2.json train_stage2;;
    pmc) bash tools/pmc_round.sh ${TAG}_pmc > $OUT/pmc_round.log 2>&1; tail -20 $OUT/pmc_round.log
         bash tools/pmc_mfma.sh ${TAG}_pmc_mfma > $OUT/pmc_mfma.log 2>&1; tail -30 $OUT/pmc_mfma.log;;
    *) echo "unknown step $s";;
  esac
done
echo "== $(( $(date +%s) - t0 )) s: done"
