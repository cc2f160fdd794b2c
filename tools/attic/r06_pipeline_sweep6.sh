export TMPDIR=/tmp
run() { echo "== $*"; timeout 300 python bench.py --no-cpu-baseline --no-bs1 --no-secondary "$@" 2>/dev/null | python tools/bench_brief.py /dev/stdin | head -1 | cut -c1-50; }
for rep in 1 2 3 4 5 6 7; do
run --steps 20
run --steps 20 --inflight 3
done
for rep in 1 2 3; do
run --steps 100
run --steps 100 --inflight 3
done
