#!/bin/bash
# Round-5 evidence session (repo root, on the GPU box): bash tools/attic/r05_final.sh [tag]
TAG=${1:-r05f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8 TMPDIR=/tmp
t0=$(date +%s)
bash tools/attic/gpu_r4.sh $TAG smoke bench20 bench100 ktimes layers head prof
echo "== $(( $(date +%s) - t0 )) s: workloads"
for W in mshp224 seg513 det800x1216 fp_input; do
  timeout 600 python bench.py --workload $W --steps 40 --no-cpu-baseline > $OUT/bench_$W.json 2> $OUT/bench_$W.err
  python - <<EOF
import json
d = json.loads(open('$OUT/bench_$W.json').read().strip().splitlines()[-1])
print('$W', round(d['value'], 1), d['unit'], round(d['ms_per_step'], 3), 'ms/step', d.get('config', {}).get('pipeline'))
EOF
done
echo "== $(( $(date +%s) - t0 )) s: training"
timeout 600 python bench.py --mode train --steps 20 --warmup 5 > $OUT/train_stage1.json 2> $OUT/train_stage1.err
timeout 600 python bench.py --mode train --stage 2 --steps 20 --warmup 5 > $OUT/train_stage2.json 2> $OUT/train_stage2.err
for f in train_stage1 train_stage2; do python - <<EOF
import json
d = json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1])
print('$f', round(d['value']), 'images/s', round(d['ms_per_step'], 2), 'ms/step')
EOF
done
timeout 600 python tools/train_tags.py > $OUT/train_tags.txt 2>&1; grep -A12 "^phases" $OUT/train_tags.txt
timeout 300 python tools/gdn_gemm_times.py > $OUT/gdn_gemm_times.txt 2>&1; grep -E "C =|rows|PRE|POST|forward GDN" $OUT/gdn_gemm_times.txt
timeout 300 python tools/wgrad_times.py > $OUT/wgrad_times.txt 2>&1; tail -6 $OUT/wgrad_times.txt
bash tools/train_prof.sh $TAG > $OUT/train_prof.log 2>&1; tail -12 $OUT/train_prof.log
echo "== $(( $(date +%s) - t0 )) s: PMC"
bash tools/attic/gpu_r4.sh $TAG pmc
bash tools/pmc_workload.sh $TAG mshp224 seg513 det800x1216 2>&1 | tail -12
echo "== $(( $(date +%s) - t0 )) s: done"
