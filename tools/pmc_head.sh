#!/bin/bash
# MFMA utilisation of the task head's kernels from PMC counters (as tools/pmc_mfma.sh, over tools/head_times.py):
#   bash tools/pmc_head.sh <tag>     ->  gpurun_out/<tag>/head_mfma_busy.txt
TAG=${1:-r02_pmc_head}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export GPU_MAX_HW_QUEUES=10
export TMPDIR=/tmp; ROOT=$(pwd)
i=0
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $ROOT/$OUT/g$i -o pmc -- python3 $ROOT/tools/head_times.py > $ROOT/$OUT/g$i.log 2>&1)
done
# (GFLOP per launch at bs 256: 3x3 conv2 layers 59.2 (x 208/196 issued); kernel names are shared by several layers: medians)
python3 tools/pmc_mfma_parse.py $OUT "win28|Geo<28|59.2" "win14|Geo<14|59.2" "win7|Geo<7|59.2" "kres1024|KresShape<1024|26.3" \
  "stream512x64x256res|conv1x1_stream_kernel<512, 64, 256, true|26.3" "stream256x128x256res|conv1x1_stream_kernel<256, 128, 256, true|26.3" \
  "stream128res|conv1x1_stream_kernel<128, 128, 256, true|26.3" "stream512x64x128|conv1x1_stream_kernel<512, 64, 128, false|26.3" \
  "generic128|Cfg<128, 128, 2, 2, false|55" "s2win28|GeoS2<28|59.2" "s2win14|GeoS2<14|59.2" "s2win7|GeoS2<7|59.2" > $OUT/head_mfma_busy.txt 2>&1
cat $OUT/head_mfma_busy.txt
