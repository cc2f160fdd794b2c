#!/bin/bash
# MFMA utilisation of the reference-precision (f32 operand) encoder kernels from PMC counters:
#   bash tools/pmc_f32.sh <tag>     ->  gpurun_out/<tag>/mfma_busy.txt  (copy to profiles/)
TAG=${1:-r04_pmc_f32}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp; ROOT=$(pwd)
i=0
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $ROOT/$OUT/g$i -o pmc -- python3 $ROOT/tools/f32_times.py > $ROOT/$OUT/g$i.log 2>&1)
done
python3 tools/pmc_mfma_parse.py $OUT 'enc.conv0+gdn1 f32 (persistent)|conv0_gdn_f32_persist_kernel|105.4' 'enc.conv0+gdn1 f32 (tile form)|conv_f32_kernel<6, 2, true>|105.4' 'enc.conv2+gdn3 f32|conv_f32_kernel<3, 2, true>|188.7' 'enc.conv4 f32|conv_f32_kernel<2, 2, false>|7.1' > $OUT/mfma_busy.txt 2>&1
cat $OUT/mfma_busy.txt
python3 - <<PY
import csv,glob,collections
res=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/g3/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'conv_f32' in r['Kernel_Name']: res[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in res.items():
    print(k,{c:sorted(v)[len(v)//2] for c,v in d.items()})
PY
# HBM-side traffic (separate passes, corrected as tools/pmc_parse.py does: FETCH_SIZE x 2)
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $ROOT/$OUT/pmc_$C -o pmc -- python3 $ROOT/tools/f32_times.py > $ROOT/$OUT/pmc_$C.log 2>&1)
done
python3 - <<PY | tee $OUT/traffic.txt
import csv,glob,collections
res=collections.defaultdict(lambda: collections.defaultdict(list))
for c in ('FETCH_SIZE','WRITE_SIZE'):
    for f in glob.glob('$OUT/pmc_'+c+'/**/*counter_collection.csv',recursive=True):
        for r in csv.DictReader(open(f)):
            if 'f32' in r['Kernel_Name'] and r['Counter_Name']==c: res[r['Kernel_Name'][:80]][c].append(float(r['Counter_Value']))
for k,d in res.items():
    f=sorted(d['FETCH_SIZE'])[len(d['FETCH_SIZE'])//2]*2048; w=sorted(d['WRITE_SIZE'])[len(d['WRITE_SIZE'])//2]*1024
    print('%-82s fetch %8.1f MB  write %8.1f MB'%(k,f/1e6,w/1e6))
PY
