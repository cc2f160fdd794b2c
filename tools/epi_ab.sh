#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_bottleneck.py -x -q 2>&1 | tail -4
