#!/bin/bash
# tiles per workgroup of the persistent decoder kernel: stand-alone (layer_times) and inside the pipelined bench, same box
export GPU_MAX_HW_QUEUES=8
for r in 1 2; do
for c in 2 3 4 6 100; do
  echo "-- SC2_CONV_CHUNK=$c"
  SC2_CONV_CHUNK=$c timeout 300 python tools/layer_times.py --bs 256 2>&1 | grep -E "dec.conv2\+igdn256|dec.conv4|synthesis"
  SC2_CONV_CHUNK=$c timeout 300 python bench.py --no-cpu-baseline --no-bs1 --steps 40 --warmup 8 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r['kernels_ms']
print('   bench K=40: %.0f img/s  %.3f ms/step  dec.conv2+igdn %.3f  dec.conv4 %.3f  fwd %.3f' % (r['value'], r['ms_per_step'], k['dec.conv2+dec.igdn3'], k['dec.conv4'], r['bottleneck_forward']['ms_per_batch_sum_of_mfma_kernels']))"
done; done
