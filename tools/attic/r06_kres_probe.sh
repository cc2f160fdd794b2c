export TMPDIR=/tmp
for L in new kres_l2; do
if [ $L = new ]; then unset SC2_LIB; else export SC2_LIB=tools/variants/lib_$L.so; fi
echo "== $L"; python tools/clock_probe.py --head-layers 2>&1 | grep -v amdgpu.ids | grep "c1\|launch"
done
