export TMPDIR=/tmp
sec() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d.get('secondary') or {}
print(round(d['value']), round(d['f32_mode']['images_per_s']) if d.get('f32_mode') else None, {k:(round(v['value']) if isinstance(v,dict) and 'value' in v else v) for k,v in s.items()})"; }
echo "== default"; timeout 600 python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | sec
echo "== --inflight 4"; timeout 600 python bench.py --steps 20 --no-cpu-baseline --inflight 4 2>/dev/null | sec
echo "== --no-bs1"; timeout 600 python bench.py --steps 20 --no-cpu-baseline --no-bs1 2>/dev/null | sec
echo "== default K=100"; timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | sec
for W in mshp224 seg513 det800x1216 fp_input; do echo "== stand-alone $W"; timeout 300 python bench.py --workload $W --steps 40 --no-cpu-baseline 2>/dev/null | sec; done
