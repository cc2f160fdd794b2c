#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for V in 0 1 0 1; do SC2_CONV_STREAM=$V timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stream', $V, round(d['value']), round(d['ms_per_step'],2), d['bpp'], round(d['bottleneck_forward']['ms_per_batch_sum_of_mfma_kernels'],3))"; done
