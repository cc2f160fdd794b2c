export TMPDIR=/tmp
for r in 1 2 3; do
echo "== k_times default"; python tools/k_times.py --only enc.conv2 --iters 60 2>&1 | grep "enc\.conv2"
for v in $T; do echo "== k_times $v"; SC2_LIB=tools/variants/lib_$v.so python tools/k_times.py --only enc.conv2 --iters 60 2>&1 | grep "enc\.conv2"; done
done
