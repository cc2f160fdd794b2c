export TMPDIR=/tmp
for K in 40 100 200; do echo "== K=$K"; timeout 300 python bench.py --workload mshp224 --steps $K --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']), round(d['ms_per_step'],3), d['config'].get('pipeline'))"; done
