#!/usr/bin/env python
"""Yardstick for the convolution family's roofline fraction: what the vendor library's PLAIN fp16 GEMM (torch.mm -> hipBLASLt; operands
already fp16 in memory, no cut, no convolution addressing, one product per fp32 product) reaches on this box at the GEMM shapes of the
family's large launches.  The fp16 two-piece form issues three such products per fp32 product: a library-grade kernel with the same
operands would deliver (library TFLOP/s) / 3 fp32-equivalent.  Lab only: nothing in the product calls a library GEMM.
    python3 tools/lab/gemm_yardstick.py [reps]
"""
import sys

import torch

DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SHAPES = [  # name, M (pixels of six images / nine frames), K, N
    ("feat_conv_3x3 x6", 14364, 18432, 1024),
    ("res5 dcn x6", 14364, 4608, 512),
    ("res5 conv1 x6", 14364, 2048, 512),
    ("res5 conv3 x6", 14364, 512, 2048),
    ("res4 conv2 x6", 14364, 2304, 256),
    ("res4 conv1 x6", 14364, 1024, 256),
    ("res4 conv3 x6", 14364, 256, 1024),
    ("res3 conv2 x6", 56250, 1152, 128),
    ("res2 conv3 x6", 225000, 64, 256),
    ("small feat x9", 21546, 2304, 1024),
    ("rfcn maps x9", 21546, 1024, 1920),
]
g = torch.Generator(device=DEV).manual_seed(0)
print("# torch %s; fp16 x fp16 -> fp16 output, fp32 accumulate (hipBLASLt); peak 2500 TFLOP/s dense fp16; %d repetitions per shape" % (torch.__version__, reps))
print("%-20s %8s %8s %8s   %9s %10s %8s   %s" % ("shape", "M", "K", "N", "us", "TFLOP/s", "of peak", "/3 = fp32-equivalent TFLOP/s (of 833)"))
for name, M, K, N in SHAPES:
    a = torch.randn((M, K), device=DEV, generator=g).half()
    b = torch.randn((K, N), device=DEV, generator=g).half()
    for _ in range(3):
        torch.mm(a, b)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.mm(a, b)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / reps
    tf = 2.0 * M * K * N / us / 1e6
    print("%-20s %8d %8d %8d   %9.1f %10.1f %8.3f   %7.1f (%.3f)" % (name, M, K, N, us, tf, tf / 2500.0, tf / 3, tf / 3 / 833.3))
