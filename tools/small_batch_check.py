#!/usr/bin/env python
"""Small-net feature of 9 frames: nine batch-1 passes vs one batch-9 pass (hipGraph replay)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lsfa_amd import tuning
from lsfa_amd.config.config import lsfa_test_config
from lsfa_amd.symbols import params as P
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
tuning.enable()
torch.backends.cudnn.benchmark = True
dev = 'cuda:0'
cfg = lsfa_test_config(key_frame_interval=10)
arg, aux = P.init_params(cfg, seed=0)
cur = resnet_v1_101_flownet_rfcn(cfg).get_cur_test_symbol(cfg).bind(arg, aux, dev)
x1 = torch.rand(1, 3, 600, 1000, device=dev) * 255

def graph_time(fn, n=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

print('batch 1: %.1f us per frame' % graph_time(lambda: cur.small_net_feature(x1)))
for nb in (3, 9):
    xb = torch.rand(nb, 3, 600, 1000, device=dev) * 255
    tb = graph_time(lambda: cur.small_net_feature(xb))
    print('batch %d: %.1f us = %.1f us per frame' % (nb, tb, tb / nb))
    a = cur.small_net_feature(xb)[1:2]; b = cur.small_net_feature(xb[1:2])
    print('   max |batched - single| = %.3g (max |x| %.3g)' % (float((a - b).abs().max()), float(b.abs().max())))
