#!/bin/bash
# A variant build of the library for A/B runs: conv.hip recompiled with extra flags, every other object taken from the product build.
#   tools/lab/build_variant.sh NAME "-DLSFA_RING_B_AUX=16"     ->  tools/lab/_build/var/liblsfa_hip_NAME.so   (select it with LSFA_HIP_LIBRARY)
# Built HERE (hipcc cross-compiles without a GPU); the .so travels to the GPU box with the snapshot.
set -eu
cd "$(dirname "$0")/../.."
NAME=$1; shift
EXTRA="$*"
B=tools/lab/_build/var
mkdir -p $B
python -m lsfa_amd.build >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -I include -I lsfa_amd/csrc \
  -fno-slp-vectorize -fno-vectorize $EXTRA -c lsfa_amd/csrc/conv.hip -o $B/conv_$NAME.o
OBJS=$(ls lsfa_amd/csrc/_obj/*.o | grep -v '/conv\.o$')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/liblsfa_hip_$NAME.so $OBJS $B/conv_$NAME.o -Wl,-rpath,/opt/rocm/lib
echo $B/liblsfa_hip_$NAME.so
