#!/usr/bin/env python
"""Launch only the cur-path warp kernel N times (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsfa_amd import hip
dev = 'cuda:0'
C, H, W = 1024, 38, 63
feat, add = torch.randn(1, C, H, W, device=dev), torch.randn(1, C, H, W, device=dev)
flow = torch.randn(1, 2, H, W, device=dev) * 0.3 + 1.5
res, res_w, res_b = torch.randn(1, 3, H, W, device=dev), torch.randn(C, 3, device=dev) * 0.01, torch.randn(C, device=dev) * 0.01
out = torch.empty_like(feat)
big = torch.empty(160 * 1024 * 1024, device=dev)   # 640 MB: flushes L2 + Infinity Cache between launches
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
flush = len(sys.argv) > 2 and sys.argv[2] == 'cold'
for i in range(n):
    if flush:
        big.fill_(float(i))
    hip.warp_bilinear(feat, flow, add=add, res=res, res_w=res_w, res_b=res_b, out=out)
torch.cuda.synchronize()
print('done', n, 'cold' if flush else 'warm')
