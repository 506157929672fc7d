#!/usr/bin/env python
"""For a rocprofv3 kernel trace: warp / aggregate next to plain streaming kernels of the same bytes."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lsfa_amd import hip
dev = 'cuda:0'
C, H, W = 1024, 38, 63
feat, feat2 = torch.randn(1, C, H, W, device=dev), torch.randn(1, C, H, W, device=dev)
flow = torch.randn(1, 2, H, W, device=dev) * 0.3 + 1.5
res, res_w, res_b = torch.randn(1, 3, H, W, device=dev), torch.randn(C, 3, device=dev) * 0.01, torch.randn(C, device=dev) * 0.01
lg = torch.randn(2, 1, H, W, device=dev)
o = torch.empty_like(feat)
for _ in range(60):
    hip.warp_bilinear(feat, flow, add=feat2, res=res, res_w=res_w, res_b=res_b, out=o)
    hip.aggregate_softmax2(feat, feat2, lg, out=o)
    torch.add(feat, feat2, out=o)
    o.copy_(feat)
    hip.scale_shift_relu(feat, res_b, res_b, True, out=o)
torch.cuda.synchronize()
