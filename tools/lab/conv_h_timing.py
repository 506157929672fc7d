#!/usr/bin/env python
"""fp16 two-piece form (lsfa_conv_split_h_fwd, three matrix instructions per product) against the bf16 three-piece form
(lsfa_conv_split_fwd, six) on the network's large convolutions: time (events around 30 back-to-back calls incl. reduce pass; the
fp16 form with and without its amax pass) and error against float64."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lsfa_amd import hip
dev = 'cuda:0'
g = torch.Generator(device=dev).manual_seed(0)


def t(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


shapes = [('feat 3x3 d6 2048->1024', (38, 63, 2048, 1024, 3, 6, 6)), ('fuse 3x3 256->1024', (38, 63, 256, 1024, 3, 1, 1)),
          ('dcn gemm 4608->512', (38, 63, 4608, 512, 1, 0, 1)), ('res4 conv2 256->256', (38, 63, 256, 256, 3, 1, 1)),
          ('res5 conv1 2048->512', (38, 63, 2048, 512, 1, 0, 1)), ('res5 conv3 512->2048', (38, 63, 512, 2048, 1, 0, 1)),
          ('res4 conv1 1024->256', (38, 63, 1024, 256, 1, 0, 1)), ('res4 conv3 256->1024', (38, 63, 256, 1024, 1, 0, 1)),
          ('rfcn 512->1920', (38, 63, 512, 1920, 1, 0, 1))]
for name, (H, W, ci, co, k, pad, dil) in shapes:
    x = torch.relu(torch.randn((1, H, W, ci), device=dev, generator=g)) * 3
    w = torch.randn((co, ci, k, k), device=dev, generator=g) * 0.01
    sw, swh = hip.SplitWeight(w), hip.SplitWeightH(w)
    am = hip.amax_partial(x)
    y6 = hip.conv_split(x, sw, None, 1, pad, dil, relu=True)
    y3 = hip.conv_split_h(x, swh, None, 1, pad, dil, act=1, amax=am)
    ref = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), padding=pad, dilation=dil)).permute(0, 2, 3, 1)
    sc = ref.abs().max().item()
    e6 = (y6.double().cpu() - ref).abs().max().item() / sc
    e3 = (y3.double().cpu() - ref).abs().max().item() / sc
    t6 = t(lambda: hip.conv_split(x, sw, None, 1, pad, dil, relu=True))
    t3 = t(lambda: hip.conv_split_h(x, swh, None, 1, pad, dil, act=1, amax=am))
    t3a = t(lambda: hip.conv_split_h(x, swh, None, 1, pad, dil, act=1))
    tl = t(lambda: torch.mm(x.view(-1, ci), w.view(co, ci).t())) if k == 1 else float('nan')
    print('%-24s six %7.1f us (err %.2e)   three %7.1f us (err %.2e)   three + amax pass %7.1f us   library GEMM %7.1f us' % (name, t6, e6, t3, e3, t3a, tl), flush=True)
