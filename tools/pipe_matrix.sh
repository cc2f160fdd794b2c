#!/bin/bash
# Pipeline-shape experiment: coder streams x ramp x max-inflight, K = 20 and K = 100 (same box, back to back).
OUT=gpurun_out/${1:-pipe}; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=8
for cfg in "2 0 24" "2 1 24" "3 1 24" "4 1 24" "2 1 16" "4 1 16"; do
  set -- $cfg
  for K in 20 100; do
    f=$OUT/i$1_r$2_m$3_k$K.json
    timeout 300 python bench.py --no-cpu-baseline --no-bs1 --steps $K --warmup 5 --inflight $1 --ramp $2 --max-inflight $3 > $f 2>/dev/null
    python - <<PY
import json
r=json.load(open("$f"))
print("inflight $1 ramp $2 maxinfl $3 K $K: %.0f img/s %.2f ms/step host %.2f lat %.0f dec.conv2 %.3f fwd %.3f enc %.1f dec %.1f" % (r['value'], r['ms_per_step'], r['host_issue_ms_per_step'], r['latency_ms_per_batch']['mean'], r['kernels_ms']['dec.conv2+dec.igdn3'], r['bottleneck_forward']['ms_per_batch_sum_of_mfma_kernels'], r['kernels_ms']['rans_encode'], r['kernels_ms']['rans_decode']))
PY
  done
done
