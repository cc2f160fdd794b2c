#!/bin/bash
# every A/B switch of the library / host layer keeps the GPU parity suite green: runs `pytest -m gpu` once per switch setting
# (the defaults are what tools/attic/gpu_round2.sh tests)
export GPU_MAX_HW_QUEUES=8
for cfg in "SC2_WIN_HALF=0" "SC2_CONV_WIN=0" "SC2_CONV1X1_WIN=0" "SC2_CONV1X1_WIN=all" "SC2_P1_HALF=1" "SC2_P1_NBUF=2" "SC2_CONV_C48=0" \
           "SC2_FC_KERNEL=0" "SC2_RANS_FUSED_DQ=0" "SC2_W2_TAIL=0" "SC2_CONV2X2_WIN=0" "SC2_CONV_STREAM=0" "SC2_CONV_KRES=0"; do
  echo "== $cfg: $(env $cfg timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1)"
done
