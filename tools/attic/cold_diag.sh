# first process on a fresh box vs later ones: where does the cold-start penalty of the K=20 line come from?
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out/cold
W=${1:-5}
for r in 1 2; do
python bench.py --gpus 1 --steps 20 --warmup $W --no-cpu-baseline --no-bs1 --diag-repeat 2 > gpurun_out/cold/r$r.json 2> gpurun_out/cold/r$r.err; echo "process $r (warmup $W)"; python tools/bench_brief.py gpurun_out/cold/r$r.json | head -1; grep diag-repeat gpurun_out/cold/r$r.err
done
