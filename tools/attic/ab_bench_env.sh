#!/bin/bash
# A/B of an environment switch on the bench line, ONE box, interleaved rounds:
#   tools/attic/ab_bench_env.sh VAR "<values>" [rounds]
VAR=$1; VALS=$2; R=${3:-2}
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/abenv
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bs1 > /dev/null 2>&1
for r in $(seq $R); do
  for v in $VALS; do
    for K in 20 100; do
      env $VAR=$v python bench.py --steps $K --warmup 5 --no-cpu-baseline --no-bs1 > gpurun_out/abenv/${VAR}_${v}_${K}_$r.json 2>/dev/null
      echo "$VAR=$v K=$K run $r: $(python tools/bench_brief.py gpurun_out/abenv/${VAR}_${v}_${K}_$r.json | head -1 | cut -c1-64)"
    done
  done
done
