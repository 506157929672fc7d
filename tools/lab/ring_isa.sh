#!/bin/bash
# usage: tools/lab/ring_isa.sh [-DISA_NT=4 -DISA_PC=2 ...]  -> tools/lab/_build/ring_isa.s (+ resource usage on stdout)
set -eu
cd "$(dirname "$0")/../.."
B=tools/lab/_build
mkdir -p $B
/opt/rocm/bin/hipcc --offload-arch=gfx950 --offload-device-only -S -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fno-vectorize \
  -Rpass-analysis=kernel-resource-usage -I include -I lsfa_amd/csrc "$@" tools/lab/ring_isa.hip -o $B/ring_isa.s 2>&1 | grep -A12 "conv_ring_kernel.h" | grep -E "VGPRs:|Spill|Occupancy|LDS Size|error|warning" || true
wc -l $B/ring_isa.s
