#!/bin/bash
# Copies the outputs of `bash tools/attic/r05_final.sh <tag>` (merged back under gpurun_out/) into profiles/ and installs the PMC traffic files.
# usage (repo root): bash tools/install_evidence.sh r05k
T=$1; O=gpurun_out/$T
for f in bench20.json bench100.json bench_kernel_stats.csv prof_bench.json k_times.txt layer_times.txt head_times.txt train_stage1.json \
         train_stage2.json train_tags.txt gdn_gemm_times.txt wgrad_times.txt bench_mshp224.json bench_seg513.json bench_det800x1216.json \
         bench_fp_input.json smoke.log gpu_tests.log; do
  [ -f $O/$f ] && cp $O/$f profiles/${T}_$f
done
cp gpurun_out/${T}_pmc/traffic_tags.json profiles/traffic.json
cp gpurun_out/${T}_traffic_workloads.json profiles/traffic_workloads.json
cp gpurun_out/${T}_pmc/traffic.txt profiles/${T}_pmc_traffic.txt
cp gpurun_out/${T}_pmc_mfma/mfma_busy.txt profiles/${T}_pmc_mfma_busy.txt
cp gpurun_out/${T}_train_kernel_stats.csv gpurun_out/${T}_train_prof_bench.json profiles/
for w in mshp224 seg513 det800x1216; do cp gpurun_out/${T}_$w/traffic.txt profiles/${T}_pmc_traffic_$w.txt; done
python - <<EOF
import json, sys
sys.path.insert(0, '.')
import sc2bench_amd as S
print('traffic.json measured on', json.load(open('profiles/traffic.json')).get('_measured_on'))
print('this tree               ', S.hip.library_fingerprint())
EOF
