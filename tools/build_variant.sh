#!/bin/bash
# builds an A/B variant of libsc2amd.so: tools/build_variant.sh <name> [--rev <git rev>] [extra hipcc flags...]
# (--rev: compile the csrc/ of that commit instead of the working tree).  Select at run time with SC2_LIB=tools/variants/lib_<name>.so
set -e
NAME=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/sc2-benchmark_amd/csrc"
if [ "$1" == "--rev" ]; then
  REV=$2; shift; shift
  TMP=$(mktemp -d)
  (cd "$ROOT" && git archive "$REV" sc2-benchmark_amd/csrc include | tar -x -C "$TMP")
  SRC="$TMP/sc2-benchmark_amd/csrc"
fi
OUT="$ROOT/tools/variants"
mkdir -p "$OUT/_obj_$NAME"
OBJS=""
cd "$SRC"
for f in *.cpp *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -D__HIP_PLATFORM_AMD__=1 -x hip "$@" -c $f -o "$OUT/_obj_$NAME/$f.o" &
  OBJS="$OBJS $OUT/_obj_$NAME/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/lib_$NAME.so" $OBJS
echo built tools/variants/lib_$NAME.so
