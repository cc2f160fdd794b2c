export TMPDIR=/tmp
run() { echo "== $*"; timeout 600 python bench.py "$@" --no-cpu-baseline --no-secondary --no-bs1 2>/dev/null | python tools/bench_brief.py /dev/stdin | head -1 | cut -c1-50; }
for rep in 1 2; do
run --steps 100 --warmup 3
run --steps 100 --warmup 20
run --steps 100 --warmup 60
run --steps 50 --warmup 3
run --steps 50 --warmup 10
run --steps 200 --warmup 3
run --steps 400 --warmup 3
done
