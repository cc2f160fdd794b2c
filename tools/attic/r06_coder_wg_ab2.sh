export TMPDIR=/tmp
for L in new wg1; do
if [ $L = wg1 ]; then export SC2_LIB=tools/variants/lib_wg1.so; else unset SC2_LIB; fi
timeout 300 python bench.py --no-cpu-baseline --no-bs1 --no-secondary --steps 100 2>/dev/null > /tmp/b_$L.json
python - <<PY
import json
d=json.loads(open('/tmp/b_$L.json').read().strip().splitlines()[-1])
print('$L', d['value'], {k:v for k,v in d.items() if 'coder' in k or 'rans' in k})
for k in ('per_kernel','kernels','stages'):
    if k in d: print(k, {kk:vv for kk,vv in d[k].items() if 'rans' in kk or 'coder' in kk or 'enc.conv2' in kk or 'dec.conv2' in kk})
PY
done
