#!/bin/bash
# A variant build of the library for A/B runs: the named sources (default conv.hip) recompiled with extra flags, every other object taken from
# the product build.
#   tools/lab/build_variant.sh NAME "-DLSFA_RING_B_AUX=16" [conv.hip ...]   ->  tools/lab/_build/var/liblsfa_hip_NAME.so   (select it with LSFA_HIP_LIBRARY)
# Built HERE (hipcc cross-compiles without a GPU); the .so travels to the GPU box with the snapshot.
set -eu
cd "$(dirname "$0")/../.."
NAME=$1; EXTRA="$2"; shift 2
SRCS="${*:-conv.hip}"
B=tools/lab/_build/var
mkdir -p $B
python -m lsfa_amd.build >/dev/null
OBJS=$(ls lsfa_amd/csrc/_obj/*.o)
for src in $SRCS; do
  o=$B/${src%.hip}_$NAME.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -I include -I lsfa_amd/csrc \
    -fno-slp-vectorize -fno-vectorize $EXTRA -c lsfa_amd/csrc/$src -o $o
  OBJS=$(echo "$OBJS" | grep -v "/${src%.hip}\.o$")
  OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/liblsfa_hip_$NAME.so $OBJS -Wl,-rpath,/opt/rocm/lib
echo $B/liblsfa_hip_$NAME.so
