#!/bin/bash
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for r in 1 2; do for V in 0 1; do echo "== SC2_B_TILE=$V"; SC2_B_TILE=$V timeout 300 python tools/layer_times.py --bs 256 2>&1 | grep -E "enc\.conv|dec\.conv2|dec.conv4|igdn256|analysis|synthesis|head\(hip"; done; done
