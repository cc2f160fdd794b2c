export TMPDIR=/tmp
V=${V:-sched0}
python -m pytest tests/test_gpu_kernels.py -q -x -k "conv2_gdn48 or persistent" 2>&1 | tail -2
echo "== digest new"; python tools/attic/enc2_digest.py 2>&1 | grep -v amdgpu.ids
for r in 1 2 3 4; do
echo "== k_times new"; python tools/k_times.py --only enc.conv2 --iters 60 2>&1 | grep "enc\.conv2"
for v in $V; do echo "== k_times $v"; SC2_LIB=tools/variants/lib_$v.so python tools/k_times.py --only enc.conv2 --iters 60 2>&1 | grep "enc\.conv2"; done
done
echo "== stamps new"; SC2_LIB=tools/variants/lib_st_new.so SC2_ENC2_STAMPS=/tmp/st2.bin python tools/enc2_stamps.py 2>&1 | grep -v amdgpu.ids | grep "wg 0 wave [01]"
