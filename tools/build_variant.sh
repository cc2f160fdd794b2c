#!/bin/bash
# builds an A/B variant of libsc2amd.so with extra -D flags: tools/build_variant.sh <name> <flags...>
set -e
NAME=$1; shift
cd "$(dirname "$0")/../sc2-benchmark_amd/csrc"
mkdir -p ../../tools/variants/_obj_$NAME
OBJS=""
for f in abi.cpp cdf_host.cpp layout.hip conv_igemm.hip conv_wgrad.hip gdn_bwd.hip entropy.hip rans.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -x hip "$@" -c $f -o ../../tools/variants/_obj_$NAME/$f.o &
  OBJS="$OBJS ../../tools/variants/_obj_$NAME/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/variants/lib_$NAME.so $OBJS
echo built tools/variants/lib_$NAME.so
