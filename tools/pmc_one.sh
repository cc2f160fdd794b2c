#!/bin/bash
# PMC counters for the bottleneck kernels (one pass per counter group): bash tools/pmc_one.sh <tag> "<kernel name regex>"
TAG=${1:-pmc}; PAT=${2:-conv2x2}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; ROOT=$(pwd)
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --kernel-trace --pmc $G --output-format csv -d $ROOT/$OUT/g$i -o pmc -- python3 $ROOT/tools/layer_times.py --bs 256 --iters 3 > $ROOT/$OUT/g$i.log 2>&1)
done
python3 - <<PY
import csv, glob, collections, re
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/g*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        name = row.get('Kernel_Name') or ''
        if re.search(r'$PAT', name):
            res[name[:90]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, d in res.items():
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v); print('   {:<34} median {:>16.0f}  (n={})'.format(c, v[len(v)//2], len(v)))
PY
