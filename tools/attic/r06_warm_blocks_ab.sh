export TMPDIR=/tmp
run() { timeout 600 python bench.py --gpus 1 "$@" --no-cpu-baseline --no-secondary --no-bs1 2>/tmp/e.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']), round(d['ms_per_step'],3))"; true; }
for rep in 1 2 3 4 5; do
echo "== warm blocks K=20"; export SC2_WARM_BLOCKS=1; run --steps 20 --warmup 5
echo "== before K=20"; export SC2_WARM_BLOCKS=0; run --steps 20 --warmup 5
done
for rep in 1 2; do
echo "== warm blocks K=100"; export SC2_WARM_BLOCKS=1; run --steps 100
echo "== before K=100"; export SC2_WARM_BLOCKS=0; run --steps 100
done
