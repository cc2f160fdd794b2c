#!/bin/bash
# Memory-side counters of the f32 encoder kernels (one --pmc pass per group):  bash tools/pmc_f32_mem.sh <tag>
TAG=${1:-r04_pmc_f32_mem}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; ROOT=$(pwd)
i=0
for G in "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_READ_sum" \
         "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_BUSY_sum TCC_CYCLE_sum TCC_EA0_RDREQ_DRAM_sum" \
         "GRBM_GUI_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $ROOT/$OUT/m$i -o pmc -- python3 $ROOT/tools/f32_times.py > $ROOT/$OUT/m$i.log 2>&1) || echo "pass $i failed: $(tail -2 $OUT/m$i.log)"
done
python3 - <<PY | tee $OUT/mem.txt
import csv,glob,collections
res=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/m*/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_f32' in r['Kernel_Name']: res[r['Kernel_Name'].split('conv_f32_kernel')[1][:14]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in sorted(res.items()):
    print(k)
    for c,v in sorted(d.items()): print('   %-40s %.4g'%(c,sorted(v)[len(v)//2]))
PY
