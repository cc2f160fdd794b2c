#!/bin/bash
# Round 6 quick pass: selected tests + a bench line.  Usage: bash tools/r06_q.sh <tag> "<pytest args>" [bench20|bench100|none]
TAG=${1:-r06q}; TESTS=${2:-}; BENCH=${3:-bench20}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -25 > $OUT/tests.log; cat $OUT/tests.log; fi
case $BENCH in
  bench20) timeout 900 python bench.py --steps 20 --no-secondary > $OUT/bench20.json 2> $OUT/bench20.err; python tools/bench_brief.py $OUT/bench20.json; tail -3 $OUT/bench20.err;;
  bench100) timeout 900 python bench.py --no-secondary > $OUT/bench100.json 2> $OUT/bench100.err; python tools/bench_brief.py $OUT/bench100.json; tail -3 $OUT/bench100.err;;
esac
python - <<PY
import json
for f in ('bench20','bench100'):
    try:
        r=json.loads([l for l in open('$OUT/%s.json'%f) if l.startswith('{')][-1])
        print(f, 'bs1_eval', json.dumps(r.get('bs1_eval')))
        print(f, 'roofline', {k:r['roofline'].get(k) for k in ('frac','frac_steady','kernel_ms')}, 'fwd', {k:r['bottleneck_forward'].get(k) for k in ('frac_of_mfma_peak','frac_8p3418')}, 'solo', {k:r['bottleneck_forward']['stand_alone'].get(k) for k in ('frac_of_mfma_peak','frac_8p3418','ms_per_batch','ms_per_batch_8p3418')})
    except Exception as e: pass
PY
