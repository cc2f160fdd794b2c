export TMPDIR=/tmp
run() { echo "== $*"; timeout 300 python bench.py --no-cpu-baseline --no-bs1 --no-secondary "$@" 2>/dev/null | python tools/bench_brief.py /dev/stdin | head -1; }
for rep in 1 2; do
run --steps 100
run --steps 100 --coder-group 16
run --steps 100 --inflight 6
run --steps 100 --max-inflight 40
run --steps 100 --coder-group 4 --inflight 6
run --steps 20
run --steps 20 --coder-group 4
run --steps 20 --ramp 0 --coder-group 4
done
