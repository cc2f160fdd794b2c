#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/layer_times.py --bs 256 2>&1 | grep -E "enc\.|analysis"
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],2), d['bpp'], d['roofline'], d['bottleneck_forward'], d['kernels_ms'])"
