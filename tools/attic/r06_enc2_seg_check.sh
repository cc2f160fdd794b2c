export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_shapes.py tests/test_gpu_dense.py -q -x -k "conv2_gdn48 or persistent or shape or seg or det or dense" 2>&1 | tail -2
echo "== digest new"; python tools/attic/enc2_digest.py 2>&1 | grep -v amdgpu.ids
echo "== digest round-5 loop"; SC2_LIB=tools/variants/lib_sched0.so python tools/attic/enc2_digest.py 2>&1 | grep -v amdgpu.ids
