export TMPDIR=/tmp
run() { echo "== $*"; timeout 300 python bench.py --no-cpu-baseline --no-bs1 --no-secondary "$@" 2>/dev/null | python tools/bench_brief.py /dev/stdin | head -1 | cut -c1-50; }
for rep in 1 2 3; do
run --steps 20
run --steps 20 --inflight 3
run --steps 20 --inflight 2
run --steps 20 --inflight 5
run --steps 40
run --steps 40 --inflight 3
done
