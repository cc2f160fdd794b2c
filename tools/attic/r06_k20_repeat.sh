export TMPDIR=/tmp GPU_MAX_HW_QUEUES=10
for i in 1 2 3 4; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python tools/bench_brief.py /dev/stdin | head -1 | cut -c1-60; done
for i in 1 2; do timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-bs1 --no-secondary 2>/dev/null | python tools/bench_brief.py /dev/stdin | head -1 | cut -c1-60; done
