"""What hipcc emitted for the library's kernels (CPU test: the code objects are disassembled, nothing runs).

No packed-fp32 VALU math (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) may appear in liblsfa_hip.so.  hipcc's SLP
vectoriser pairs adjacent scalar float operations into those instructions (with op_sel half-selects); with them,
lsfa_deform_im2col_cl_ld returned wrong values in the low halves of one 16-lane group about once per hundred launches
WHEN ANOTHER PROCESS TIME-SHARED THE GPU (never alone) — which is how two ranks on one device, and the driver's r2 run
of tests/test_multirank_gpu.py, got whole frames of different detections (profiles/r3/multirank_*.txt,
tools/diag_garbage.py).  The build therefore passes -fno-slp-vectorize; this test keeps it that way."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump not in this image")
def test_no_packed_fp32_math_in_the_library(tmp_path):
    lib = os.path.join(ROOT, "lsfa_amd", "liblsfa_hip.so")
    if not os.path.exists(lib):
        from lsfa_amd import build
        build.build_hip()
    work = tmp_path / "liblsfa_hip.so"
    shutil.copy(lib, work)
    subprocess.run([OBJDUMP, "--offloading", str(work)], check=True, capture_output=True)
    objs = glob.glob(str(work) + ".*gfx950*")
    assert objs, "no gfx950 code object found in liblsfa_hip.so"
    packed, kernels = [], 0
    for o in objs:
        asm = subprocess.run([OBJDUMP, "-d", o], check=True, capture_output=True, text=True).stdout
        kernels += len(re.findall(r"^[0-9a-f]+ <\w+>:", asm, flags=re.M))
        packed += re.findall(r"v_pk_(?:mul|add|fma)_f32[^\n]*", asm)
    assert kernels > 20
    assert not packed, "%d packed-fp32 instructions, e.g. %s" % (len(packed), packed[0])
