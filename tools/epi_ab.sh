#!/bin/bash
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "gdn512" 2>&1 | tail -8
timeout 300 python tools/layer_times.py --bs 256 2>&1 | grep -E "dec\.|synthesis"
