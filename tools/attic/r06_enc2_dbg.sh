export TMPDIR=/tmp
for r in 1 2 3; do
echo "== k_times default"; python tools/k_times.py --only enc. --iters 40 2>&1 | grep "enc\.conv0+gdn96 (nchw\|analysis"
for v in $T; do echo "== k_times $v"; SC2_LIB=tools/variants/lib_$v.so python tools/k_times.py --only enc. --iters 40 2>&1 | grep "enc\.conv0+gdn96 (nchw\|analysis"; done
done
