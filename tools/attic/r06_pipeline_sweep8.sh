export TMPDIR=/tmp
run() { echo "== Q=$GPU_MAX_HW_QUEUES $*"; timeout 300 python bench.py --no-cpu-baseline --no-bs1 --no-secondary "$@" 2>/dev/null | python tools/bench_brief.py /dev/stdin | head -1 | cut -c1-50; }
for rep in 1 2 3 4; do
for q in 7 8 10 12; do
export GPU_MAX_HW_QUEUES=$q
run --steps 20
done
done
for rep in 1 2; do
for q in 8 10; do
export GPU_MAX_HW_QUEUES=$q
run --steps 100
done
done
