# conv2_gdn48: per-tap stamps of the round-5 slab loop and of the round-6 one, the launch with every patch row in L2, the A/B of the two loops
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "# conv2_gdn48 (enc.conv2 + enc.gdn3), 256 x 110 x 112 x 96 -> 256 x 53 x 56 x 48; cycles of s_memtime, means over 8 units of workgroup 0, waves 0 and 1"
echo "# (a stamped build drains the LDS queue at every stamp: its taps are ~60 cycles longer than the unstamped build's)"
echo "== round-5 loop (-DSC2_ENC2_SCHED=0 -DSC2_ENC2_W6=0), per tap"; SC2_LIB=tools/variants/lib_st_old.so SC2_ENC2_STAMPS=/tmp/st2.bin python tools/enc2_stamps.py 2>&1 | grep "wg 0 wave [01]"
echo "== round-6 loop, per tap"; SC2_LIB=tools/variants/lib_st_new.so SC2_ENC2_STAMPS=/tmp/st2.bin python tools/enc2_stamps.py 2>&1 | grep "wg 0 wave [01]"
for r in 1 2 3; do
echo "== launch, round-6 loop"; python tools/k_times.py --only enc.conv2 --iters 60 2>&1 | grep "enc\.conv2"
echo "== launch, round-5 loop (-DSC2_ENC2_SCHED=0 -DSC2_ENC2_W6=0)"; SC2_LIB=tools/variants/lib_sched0.so python tools/k_times.py --only enc.conv2 --iters 60 2>&1 | grep "enc\.conv2"
echo "== launch, every unit fetches the rows of unit 0 (-DSC2_ENC2_DBG=8: wrong results, every piece an L2 hit)"; SC2_LIB=tools/variants/lib_l2hit.so python tools/k_times.py --only enc.conv2 --iters 60 2>&1 | grep "enc\.conv2"
done
} > gpurun_out/r06_enc2_taps.txt 2>&1
cat gpurun_out/r06_enc2_taps.txt
