export TMPDIR=/tmp
run() { echo "== $*"; timeout 600 python bench.py "$@" 2>/tmp/err.txt | python tools/bench_brief.py /dev/stdin | head -1 | cut -c1-70; tail -2 /tmp/err.txt | grep -v amdgpu.ids; }
run --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline --no-secondary --no-bs1
run --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-bs1
run --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --no-secondary --no-bs1
run --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline --no-secondary --no-bs1
run --gpus 1 --steps 7 --warmup 0
run --gpus 1 --steps 200 --warmup 20 --no-cpu-baseline --no-secondary --no-bs1
