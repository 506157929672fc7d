#!/usr/bin/env python
"""How much the non-key frames of a segment would gain from running as ONE batch (they depend on the key frame only, not on each
other): the non-key forward replayed as a hipGraph at batch 1, 3 and 9 (every image with its own key feature, as `--clips B` runs)."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lsfa_amd.config.config import lsfa_test_config
from lsfa_amd.symbols import params as P
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
from lsfa_amd import tuning
tuning.enable()
dev = 'cuda:0'
H, W = 600, 1000
cfg = lsfa_test_config(key_frame_interval=10)
arg, aux = P.init_params(cfg, seed=0)
net = resnet_v1_101_flownet_rfcn(cfg)
cur = net.get_cur_test_symbol(cfg).bind(arg, aux, dev)


def graph_time(fn, n=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for B in (1, 3, 9):
    data = torch.rand(B, 3, H, W, device=dev) * 255
    im_info = torch.tensor([[H, W, 1.0]] * B, device=dev)
    feat = torch.randn(B, 1024, 38, 63, device=dev)
    mv = torch.randn(B, 2, 38, 63, device=dev)
    res = torch.randn(B, 3, 38, 63, device=dev)
    t_small = graph_time(lambda: cur.small_net_feature(data))
    t_all = graph_time(lambda: cur.forward(data=data, im_info=im_info, feat_key=feat, motion_vector=mv, res_diff=res))
    print('batch %d: small net feature %8.1f us (%6.1f per frame)   whole non-key forward %8.1f us (%6.1f per frame)'
          % (B, t_small, t_small / B, t_all, t_all / B), flush=True)
