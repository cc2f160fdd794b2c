#!/bin/bash
# Development aid: share of the store epilogue / the K loop in each conv launch (SC2_CONV_DEBUG: 1 = no epilogue, 2 = no K loop)
for D in 0 1 2; do
  echo "== SC2_CONV_DEBUG=$D"
  SC2_CONV_DEBUG=$D python tools/layer_times.py --bs 256 2>&1 | grep -E "enc\.|dec\.|analysis|synthesis|head\(hip"
done
