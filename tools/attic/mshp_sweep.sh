#!/bin/bash
# coder-group / coder-stream / waves-per-workgroup sweep of the pipelined MSHP workload (bench.py --workload mshp224)
OUT=gpurun_out/${1:-r05j}; mkdir -p $OUT
for W in 1 2 4; do for G in 2 4 8; do for C in 3 4 6; do
  timeout 300 python bench.py --workload mshp224 --steps 40 --warmup 3 --no-cpu-baseline --coder-group $G --inflight $C --policy rans_ragged2_waves=$W 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('waves $W G $G streams $C :', round(d['value']), 'img/s', round(d['ms_per_step'],2), 'ms/step', 'dec.idx', d['rans']['rans_decode.indexed']['ms_per_launch'], 'enc.idx', d['rans']['rans_encode.indexed']['ms_per_launch'])
except Exception as e: print('waves $W G $G streams $C : failed', e)"
done; done; done | tee $OUT/mshp_sweep.txt
