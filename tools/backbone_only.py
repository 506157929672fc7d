#!/usr/bin/env python
"""The key frame's backbone (or `flownet` as 2nd argument) alone, eagerly, many times (for rocprofv3 kernel traces: the second half
of the trace is steady state).  3rd argument: images per pass (the batched pipeline's key groups: 3)."""
import sys
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lsfa_amd.config.config import lsfa_test_config
from lsfa_amd.symbols import params as P
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
dev = 'cuda:0'
cfg = lsfa_test_config(key_frame_interval=10)
arg, aux = P.init_params(cfg, seed=0)
key = resnet_v1_101_flownet_rfcn(cfg).get_key_test_symbol(cfg).bind(arg, aux, dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
what = sys.argv[2] if len(sys.argv) > 2 else 'backbone'
G = int(sys.argv[3]) if len(sys.argv) > 3 else 1
data = torch.rand(G, 3, 600, 1000, device=dev) * 255
data2 = torch.rand(G, 3, 600, 1000, device=dev) * 255
with torch.no_grad():
    for _ in range(n):
        key._backbone(data) if what == 'backbone' else key._flownet(data, data2)
torch.cuda.synchronize()
