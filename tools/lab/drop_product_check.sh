#!/bin/bash
# Does the float64-anchored parity criterion notice a degraded convolution?  ("the same test FAILS if mma3h drops a
# second product".)  Builds a copy of the library whose fp16 two-piece product drops the hi*lo term as well (hi*hi + lo*hi only:
# weights effectively rounded to 11 bits), runs tests/test_parity_fullres_gpu.py against it through LSFA_HIP_LIBRARY and expects
# the test to FAIL; then runs it against the real library and expects it to pass.  Lab only: the product header is not touched.
set -u
cd "$(dirname "$0")/../.."
B=tools/lab/_build/drop
rm -rf $B && mkdir -p $B/csrc && cp lsfa_amd/csrc/*.hip lsfa_amd/csrc/*.h $B/csrc/
python - $B/csrc/conv_split_kernel.h <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
old = "  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(a.hi), as_h(blo), acc, 0, 0, 0);\n"
assert s.count(old) == 1
open(p, 'w').write(s.replace(old, "  /* dropped for the lab check: hi * lo */\n"))
PY
OBJS=""
for f in $B/csrc/*.hip; do
  o=${f%.hip}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I include -I $B/csrc -fno-slp-vectorize -fno-vectorize -c $f -o $o || exit 2
  OBJS="$OBJS $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/liblsfa_hip_drop.so $OBJS -Wl,-rpath,/opt/rocm/lib || exit 2
echo "== degraded library (hi*lo dropped): the parity test must FAIL"
LSFA_HIP_LIBRARY=$PWD/$B/liblsfa_hip_drop.so python -m pytest tests/test_parity_fullres_gpu.py -x -q 2>&1 | tail -4
rc_bad=${PIPESTATUS[0]}
echo "== real library: the parity test must PASS"
python -m pytest tests/test_parity_fullres_gpu.py -x -q 2>&1 | tail -2
rc_good=${PIPESTATUS[0]}
echo "degraded rc=$rc_bad (want != 0)   real rc=$rc_good (want 0)"
[ "$rc_bad" -ne 0 ] && [ "$rc_good" -eq 0 ]
