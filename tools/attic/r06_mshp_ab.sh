export TMPDIR=/tmp
for rep in 1 2; do
for L in new sched0; do
if [ $L = sched0 ]; then export SC2_LIB=tools/variants/lib_sched0.so; else unset SC2_LIB; fi
echo "== $L"; timeout 300 python bench.py --workload mshp224 --steps 40 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']), d['ms_per_step'])"
done
done
