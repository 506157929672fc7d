#!/usr/bin/env python
"""The split-bf16 convolution (lsfa_conv_split_fwd) next to the fp32-MFMA one (lsfa_conv_nhwc_fused_fwd) and the
library's, on the ResNet-101 shapes at 1000x600: error against a float64 convolution and time.

    python tools/lab/conv_split_lab.py [--iters 30]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lsfa_amd import hip  # noqa: E402

SHAPES = [  # name, H, W, Cin, Cout, k, dil
    ("res4 conv2 3x3 256->256 @38x63", 38, 63, 256, 256, 3, 1),
    ("res4 conv1 1x1 1024->256 @38x63", 38, 63, 1024, 256, 1, 1),
    ("res4 conv3 1x1 256->1024 @38x63", 38, 63, 256, 1024, 1, 1),
    ("res3 conv2 3x3 128->128 @75x125", 75, 125, 128, 128, 3, 1),
    ("res3 conv1 1x1 512->128 @75x125", 75, 125, 512, 128, 1, 1),
    ("res3 conv3 1x1 128->512 @75x125", 75, 125, 128, 512, 1, 1),
    ("res2 conv2 3x3 64->64 @150x250", 150, 250, 64, 64, 3, 1),
    ("res2 conv3 1x1 64->256 @150x250", 150, 250, 64, 256, 1, 1),
    ("fuse 3x3 256->1024 @38x63", 38, 63, 256, 1024, 3, 1),
    ("res5 conv1 1x1 1024->512 @38x63", 38, 63, 1024, 512, 1, 1),
    ("res5 conv1 1x1 2048->512 @38x63", 38, 63, 2048, 512, 1, 1),
    ("res5 conv3 1x1 512->2048 @38x63", 38, 63, 512, 2048, 1, 1),
    ("res5 dcn gemm 1x1 4608->512 @38x63", 38, 63, 4608, 512, 1, 1),
    ("feat 3x3 d6 2048->1024 @38x63", 38, 63, 2048, 1024, 3, 6),
    ("nq conv 1x1 2048->256? @38x63", 38, 63, 2048, 256, 1, 1),
    ("small conv2 3x3 64->64 @38x63", 38, 63, 64, 64, 3, 1),
    ("small conv1 1x1 256->64 @38x63", 38, 63, 256, 64, 1, 1),
    ("small conv3 1x1 64->256 @38x63", 38, 63, 64, 256, 1, 1),
    ("rpn 1x1 512->64 @38x63", 38, 63, 512, 64, 1, 1),
]


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


PHASES = ["issue the 7 LDS-DMAs of chunk c+2", "A read + cut + 24 MFMAs of chunk c", "wait for chunk c+1's DMAs", "barrier"]


def stamps(dev):
    """Shader-clock stamps around the phases of one steady-state chunk (res4 conv2, 6 slices)."""
    import ctypes
    import subprocess
    import numpy as np
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "_build")
    os.makedirs(out, exist_ok=True)
    so, src = os.path.join(out, "libconv_split_lab.so"), os.path.join(here, "conv_split_lab.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-fno-fast-math", "-DLSFA_CS_STAMPS", "-I", os.path.join(ROOT, "lsfa_amd", "csrc"), src, "-o", so])
    lab = ctypes.CDLL(so)
    H, W, C, k = 38, 63, 256, 3
    x = torch.randn(1, H, W, C, device=dev)
    sw = hip.SplitWeight(torch.randn(C, C, k, k, device=dev) * 0.02)
    part = torch.empty(6, H * W, C, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    acc = np.zeros(4)
    n = 20
    for it in range(n + 3):
        st = (ctypes.c_longlong * 8)()
        assert lab.conv_split_lab_run(vp(x), vp(sw.frag), vp(part), 1, H, W, C, C, k, 6, st) == 0
        if it >= 3:
            acc += np.diff(np.array(st[:5], dtype=np.float64))
    print("one chunk of res4 conv2 (shader cycles, wave 0 of one workgroup, mean of %d launches):" % n)
    for name, cyc in zip(PHASES, acc / n):
        print("  %-42s %8.0f" % (name, cyc))
    print("  %-42s %8.0f   (24 bf16 MFMAs = 768 cycles of matrix pipe)" % ("total", acc.sum() / n))


def slices_sweep(dev, iters=30):
    """res4 conv2 and res4 conv1 through the general kernel alone (no reduce pass) for different K cuts."""
    import ctypes
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    so, src = os.path.join(here, "_build", "libconv_split_lab_plain.so"), os.path.join(here, "conv_split_lab.hip")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                           "-fno-fast-math", "-I", os.path.join(ROOT, "lsfa_amd", "csrc"), src, "-o", so])
    lab = ctypes.CDLL(so)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    for name, H, W, Cin, Cout, k in (("res4 conv2", 38, 63, 256, 256, 3), ("res4 conv1", 38, 63, 1024, 256, 1), ("res3 conv2", 75, 125, 128, 128, 3)):
        x = torch.randn(1, H, W, Cin, device=dev)
        sw = hip.SplitWeight(torch.randn(Cout, Cin, k, k, device=dev) * 0.02)
        part = torch.empty(16, H * W, Cout, device=dev)
        st = (ctypes.c_longlong * 8)()
        for S in (1, 2, 3, 4, 5, 6, 8, 9, 12):
            if (k * k * Cin // 32) // S < 4:
                continue
            fn = lambda: lab.conv_split_lab_run(vp(x), vp(sw.frag), vp(part), 1, H, W, Cin, Cout, k, S, st)
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            e1.synchronize()
            print("%-12s slices %2d: kernel %6.1f us (synchronising launches)" % (name, S, e0.elapsed_time(e1) * 1e3 / iters))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--stamps", action="store_true")
    ap.add_argument("--slices", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    if args.stamps:
        stamps(dev)
        return
    if args.slices:
        slices_sweep(dev)
        return
    torch.manual_seed(0)
    torch.backends.cudnn.benchmark = True
    for name, H, W, Cin, Cout, k, dil in SHAPES:
        x = torch.relu(torch.randn(1, Cin, H, W, device=dev)) * (torch.rand(1, Cin, 1, 1, device=dev) * 3)
        w = torch.randn(Cout, Cin, k, k, device=dev) * (1.0 / (Cin * k * k) ** 0.5)
        b = torch.randn(Cout, device=dev)
        pad = dil * (k // 2)
        ref = F.conv2d(x.double(), w.double(), b.double(), 1, pad, dil)
        scale = float(ref.abs().max())
        x_cl = x.permute(0, 2, 3, 1).contiguous()
        sw = hip.SplitWeight(w)
        wk = hip.conv_weight_kc(w)
        y_split = hip.conv_split(x_cl, sw, b, 1, pad, dil)
        y_mfma = hip.conv_nhwc(x_cl, wk, b, k, k, 1, pad, dil)
        xl = x.contiguous(memory_format=torch.channels_last)
        wl = w.contiguous(memory_format=torch.channels_last)
        y_lib = F.conv2d(xl, wl, b, 1, pad, dil)
        err = lambda y_nhwc: float((y_nhwc.permute(0, 3, 1, 2).double() - ref).abs().max()) / scale
        e_split, e_mfma = err(y_split), err(y_mfma)
        e_lib = float((y_lib.double() - ref).abs().max()) / scale
        rms = lambda y_nhwc: float(((y_nhwc.permute(0, 3, 1, 2).double() - ref) ** 2).mean().sqrt()) / scale
        out = torch.empty_like(y_split)
        hip.lib().lsfa_conv_split_set_variant(2)          # 128 x 128 workgroup tiles wherever Cout allows (r3)
        y_wide = hip.conv_split(x_cl, sw, b, 1, pad, dil)
        e_wide = err(y_wide)
        t_wide = timeit(lambda: hip.conv_split(x_cl, sw, b, 1, pad, dil, out=out), args.iters)
        hip.lib().lsfa_conv_split_set_variant(1)          # the r2 kernel: 128 x 64 tiles, 2-stage ring
        t_split = timeit(lambda: hip.conv_split(x_cl, sw, b, 1, pad, dil, out=out), args.iters)
        hip.lib().lsfa_conv_split_set_variant(3)          # the plan without the 4-stage ring
        t_nodeep = timeit(lambda: hip.conv_split(x_cl, sw, b, 1, pad, dil, out=out), args.iters)
        hip.lib().lsfa_conv_split_set_variant(0)          # the launch plan as shipped
        y_plan = hip.conv_split(x_cl, sw, b, 1, pad, dil)
        e_plan = err(y_plan)
        t_plan = timeit(lambda: hip.conv_split(x_cl, sw, b, 1, pad, dil, out=out), args.iters)
        t_mfma = timeit(lambda: hip.conv_nhwc(x_cl, wk, b, k, k, 1, pad, dil, out=out), args.iters)
        t_lib = timeit(lambda: F.conv2d(xl, wl, b, 1, pad, dil), args.iters)
        if k == 1:      # what the executor runs for 1x1 convolutions: rows x (Cin, Cout) through hipBLASLt
            rows, wt = x_cl.view(-1, Cin), w.view(Cout, Cin).t().contiguous()
            t_lib = min(t_lib, timeit(lambda: torch.addmm(b, rows, wt), args.iters))
        fl = 2.0 * H * W * Cin * Cout * k * k
        print("%-36s max err/max|y|: r2 %.2e (rms %.2e)  wide %.2e  plan %.2e  fp32-mfma %.2e (rms %.2e)  library %.2e | us: r2 kernel %6.1f  "
              "wide %6.1f  plan w/o deep ring %6.1f  PLAN %6.1f  fp32-mfma %6.1f  library %6.1f | r2 %.0f plan %.0f TFLOP/s" % (
                  name, e_split, rms(y_split), e_wide, e_plan, e_mfma, rms(y_mfma), e_lib, t_split, t_wide, t_nodeep, t_plan, t_mfma, t_lib,
                  fl / t_split / 1e6, fl / t_plan / 1e6))


if __name__ == "__main__":
    main()
