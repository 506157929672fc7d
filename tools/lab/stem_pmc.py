#!/usr/bin/env python
"""One shape of the stem convolution, a few eager launches (for rocprofv3 --pmc)."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lsfa_amd import hip
dev = 'cuda:0'
N, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (6, 600, 1000)
w = hip.stem_weight_layout(torch.randn(64, 3, 7, 7, device=dev) * 0.05)
b = torch.randn(64, device=dev)
sc, sh = torch.full((3,), 0.02, device=dev), torch.full((3,), -2.0, device=dev)
x = torch.rand(N, 3, H, W, device=dev) * 255
out = hip.stem_conv(x, w, b, sc, sh)
for _ in range(4):
    hip.stem_conv(x, w, b, sc, sh, out=out)
torch.cuda.synchronize()
