export GPU_MAX_HW_QUEUES=8
for r in 1 2; do for cfg in "--back-priority 0" "--back-priority -1" "--back-priority -1 --coder-priority -1" "--max-inflight 16" "--max-inflight 32"; do for K in 20 100; do
  timeout 300 python bench.py --no-cpu-baseline --no-bs1 --steps $K --warmup 5 $cfg 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r['kernels_ms']
print('%-40s K=%-3d: %.0f img/s  %.3f ms/step  dec.conv2+igdn %.3f  fwd %.3f frac %.3f dom %.3f lat %.0f' % ('$cfg', $K, r['value'], r['ms_per_step'], k['dec.conv2+dec.igdn3'], r['bottleneck_forward']['ms_per_batch_sum_of_mfma_kernels'], r['bottleneck_forward']['frac_of_mfma_peak'], r['roofline']['frac'], r['latency_ms_per_batch']['mean']))"
done; done; done
