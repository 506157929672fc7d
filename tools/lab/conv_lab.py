#!/usr/bin/env python
"""What bounds conv_igemm_kernel<64>: the product loop next to copies with the global loads, the LDS staging and
the LDS fragment reads switched off in turn (tools/lab/conv_lab.hip), on the ResNet stage-3 conv2 shape.

    python tools/lab/conv_lab.py [--slices 1,3,9]
"""
import argparse
import ctypes
import os
import subprocess
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
MODES = {0: "product loop", 1: "no global loads in the loop", 2: "no global loads, no LDS staging / barriers",
         3: "MFMAs only (no LDS fragment reads)"}


def build():
    out = os.path.join(HERE, "_build")
    os.makedirs(out, exist_ok=True)
    so, src = os.path.join(out, "libconv_lab.so"), os.path.join(HERE, "conv_lab.hip")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-ffp-contract=off", "-fno-fast-math", src, "-o", so])
    return ctypes.CDLL(so)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--slices", default="1,3,9")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--build-only", action="store_true")
    args = ap.parse_args()
    lab = build()
    if args.build_only:
        return
    dev = torch.device("cuda:0")
    N, H, W, C = 1, 38, 63, 256
    x = torch.randn(N, H, W, C, device=dev)
    w = torch.randn(C, 9, C, device=dev) * 0.02
    part = torch.empty(9, N * H * W, C, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    flops = 2.0 * N * H * W * C * C * 9
    for slices in [int(s) for s in args.slices.split(",")]:
        wgs = ((N * H * W + 63) // 64) * (C // 64) * slices
        print("slices %d: %d workgroups (%.2f per CU)" % (slices, wgs, wgs / 256.0))
        for mode, name in MODES.items():
            def run():
                assert lab.conv_lab_run(mode, vp(x), vp(w), vp(part), N, H, W, C, C, 1, slices, None) == 0
            for _ in range(5):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / args.iters
            print("  mode %d %-45s %7.2f us  %6.1f TFLOP/s" % (mode, name, us, flops / us / 1e6))


if __name__ == "__main__":
    main()
