# does the coder's cost come from tile quantisation (launches sized exactly to 256 CUs)?  batch sizes around 256, with and without coder kernels
export GPU_MAX_HW_QUEUES=8
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-bs1 > /dev/null 2>&1
for bs in 224 240 248 256 264 272; do for d in 0 1; do
python bench.py --steps 100 --warmup 5 --bs $bs --no-cpu-baseline --no-bs1 --diag-skip-coder $d > /tmp/b.json 2>/dev/null; echo "bs $bs diag-skip-coder $d: $(python tools/bench_brief.py /tmp/b.json | head -1 | cut -c1-50)"
done; done
