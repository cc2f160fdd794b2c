#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "gdn512" 2>&1 | tail -4
timeout 300 python tools/layer_times.py --bs 256 2>&1 | grep -E "conv0\+igdn512|synthesis"
SC2_LIB=tools/variants/lib_stamps.so SC2_DEC_STAMPS=/tmp/st.bin timeout 200 python tools/dec_stamps.py 2>&1 | tail -3
