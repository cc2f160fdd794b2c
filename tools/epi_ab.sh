#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_hyperprior.py -x -q -k "rans or hyperprior or gaussian" 2>&1 | tail -6
timeout 600 python tools/hyper_times.py --bs 256 2>&1 | tail -10
