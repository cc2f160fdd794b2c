export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/abw
for r in 1 2; do
for w in 5 16; do
python bench.py --gpus 1 --steps 20 --warmup $w --no-cpu-baseline --no-bs1 > gpurun_out/abw/w${w}_$r.json 2>/dev/null; echo "warmup $w run $r"; python tools/bench_brief.py gpurun_out/abw/w${w}_$r.json | head -1
done; done
