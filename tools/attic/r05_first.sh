#!/bin/bash
# Round 5, first GPU pass: new tests first (bounded), then the headline line and the pipelined workload lines.
TAG=${1:-r05a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "hostile or conv0_gdn96 or rans" > $OUT/tests_kernels.log 2>&1; echo "kernels rc $?" >> $OUT/tests_kernels.log
timeout 1200 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q > $OUT/tests_pipeline.log 2>&1; echo "pipeline rc $?" >> $OUT/tests_pipeline.log
timeout 600 python bench.py --steps 20 > $OUT/bench20.json 2> $OUT/bench20.err
timeout 300 python tools/k_times.py > $OUT/k_times.txt 2>&1
for w in mshp224 seg513 det800x1216 fp_input; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 3 > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
tail -3 $OUT/tests_kernels.log $OUT/tests_pipeline.log
python tools/bench_brief.py $OUT/bench20.json 2>/dev/null | head -20
for w in mshp224 seg513 det800x1216 fp_input; do python - <<PY
import json
try:
    r = json.loads([l for l in open('$OUT/bench_$w.json') if l.startswith('{')][-1])
    print('$w', round(r['value'], 1), 'img/s', round(r['ms_per_step'], 2), 'ms/step', r['config']['pipeline'] if isinstance(r['config']['pipeline'], str) else r['config']['pipeline'].get('steps_per_coder_launch'))
except Exception as e:
    print('$w failed', e, open('$OUT/bench_$w.err').read()[-600:])
PY
done
