export TMPDIR=/tmp
python -m pytest tests/test_gpu_rans.py tests/test_gpu_pipeline.py tests/test_gpu_host_coder.py -q -x 2>&1 | tail -3
run() { echo "== $L $*"; timeout 300 python bench.py --no-cpu-baseline --no-bs1 --no-secondary "$@" 2>/dev/null | python tools/bench_brief.py /dev/stdin | head -1 | cut -c1-50; }
for rep in 1 2 3; do
L=new; unset SC2_LIB; run --steps 20; run --steps 100
L=wg1; export SC2_LIB=tools/variants/lib_wg1.so; run --steps 20; run --steps 100
done
