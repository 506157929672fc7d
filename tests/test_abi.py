"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports
every symbol include/lsfa_hip.h declares; the product never touches the oracle."""
import ctypes
import os
import re

import torch  # noqa: F401  (loads the HIP runtime the library binds to)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "lsfa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if n.startswith("lsfa_") or n == "_nms"))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for want in ("lsfa_psroi_pool_fwd", "lsfa_rfcn_head_fwd", "lsfa_warp_bilinear", "lsfa_aggregate_softmax2",
                 "lsfa_aggregate_cosine", "lsfa_proposal", "lsfa_nms_sorted", "_nms", "lsfa_det_postprocess",
                 "lsfa_bbox_pred_clip", "lsfa_deform_im2col", "lsfa_deform_im2col_cl", "lsfa_deform_im2col_cl_ld", "lsfa_scale_shift_relu", "lsfa_scale_shift_relu_cl", "lsfa_scale_shift_leaky", "lsfa_prof_read"):
        assert want in syms


def test_library_builds_loads_and_exports_every_declared_symbol():
    from lsfa_amd import build
    path = build.build_hip()
    lib = ctypes.CDLL(path)
    for name in declared_symbols():
        assert hasattr(lib, name), "liblsfa_hip.so does not export %s" % name
    lib.lsfa_abi_version.restype = ctypes.c_int
    assert lib.lsfa_abi_version() == 1
    lib.lsfa_op_name.restype = ctypes.c_char_p
    from lsfa_amd import hip
    assert [lib.lsfa_op_name(i).decode() for i in range(len(hip.OP_NAMES))] == hip.OP_NAMES
    # ... and the table is COMPLETE: lsfa_prof_read fills LSFA_OP_COUNT entries of buffers the binding sizes by this list (an
    # out-of-date copy, one short, let it write past them and hung bench.py: r3)
    assert lib.lsfa_op_name(len(hip.OP_NAMES)) == b"?"
    # workspace sizing is host-only arithmetic: float4 boxes + u32 keys for the 21,546 anchors (no NMS mask)
    lib.lsfa_proposal_workspace_bytes.restype = ctypes.c_size_t
    ws = lib.lsfa_proposal_workspace_bytes(1, 9, 38, 63, 6000)
    # boxes + keys of the decode kernel, then the chip-wide plan's histograms, candidate list, ranks and the
    # 6000 x 94 x 8-byte suppression mask (the reference cudaMallocs that mask on every call)
    assert 21546 * 20 + 6000 * 94 * 8 <= ws < 8 << 20
    assert 21546 * 20 <= lib.lsfa_proposal_workspace_bytes(1, 9, 38, 63, 12000) < 21546 * 20 + 1024   # single-workgroup plan
    lib.lsfa_nms_workspace_bytes.restype = ctypes.c_size_t
    assert lib.lsfa_nms_workspace_bytes(6000) >= 6000 * 94 * 8


def test_product_never_imports_the_oracle():
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "lsfa_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f)).read()
                if re.search(r"^\s*(import|from)\s+oracle\b", txt, flags=re.M) or "lsfa_oracle" in txt or "liblsfa_oracle" in txt:
                    bad.append(os.path.join(d, f))
    assert not bad, "product code references the oracle: %s" % bad


