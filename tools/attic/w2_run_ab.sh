#!/bin/bash
# A/B of the run length (tiles per workgroup) of the window-plane decoder kernels inside the pipelined bench
export GPU_MAX_HW_QUEUES=8
for r in 1 2; do for v in 0 7 4 2 1; do for K in 20 100; do
  SC2_W2_RUN=$v timeout 300 python bench.py --no-cpu-baseline --no-bs1 --steps $K --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r['kernels_ms']
print('run %-3s K=%-3d: %.0f img/s  %.3f ms/step  dec.conv2+igdn %.3f dec.conv4 %.3f dec0 %.3f enc %.3f/%.3f/%.3f  fwd %.3f frac %.3f dom %.3f' % ('$v', $K, r['value'], r['ms_per_step'], k['dec.conv2+dec.igdn3'], k['dec.conv4'], k['dec.conv0+dec.igdn1'], k['enc.conv0+enc.gdn1'], k['enc.conv2+enc.gdn3'], k['enc.conv4'], r['bottleneck_forward']['ms_per_batch_sum_of_mfma_kernels'], r['bottleneck_forward']['frac_of_mfma_peak'], r['roofline']['frac']))"
done; done; done
