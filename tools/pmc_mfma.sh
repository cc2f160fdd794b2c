#!/bin/bash
# MFMA utilisation of the bottleneck kernels from PMC counters (one --pmc pass per counter group, --kernel-trace only):
#   bash tools/pmc_mfma.sh <tag>     ->  gpurun_out/<tag>/mfma_busy.txt  (copy to profiles/)
TAG=${1:-r02_pmc}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=10
export TMPDIR=/tmp; ROOT=$(pwd)
i=0
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $ROOT/$OUT/g$i -o pmc -- python3 $ROOT/tools/layer_times.py --bs 256 --iters 3 > $ROOT/$OUT/g$i.log 2>&1)
done
python3 tools/pmc_mfma_parse.py $OUT > $OUT/mfma_busy.txt 2>&1
cat $OUT/mfma_busy.txt
