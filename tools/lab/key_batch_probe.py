#!/usr/bin/env python
"""Lab: what the image-only half of a key frame (backbone, FlowNet) costs per frame when the fronts of several key frames are
computed in one pass (N on the batch axis), each section captured as its own hipGraph and replayed alone."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lsfa_amd.config.config import lsfa_test_config
from lsfa_amd.symbols import params as P
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
dev = 'cuda:0'
H, W = 600, 1000
cfg = lsfa_test_config(key_frame_interval=10)
arg, aux = P.init_params(cfg, seed=0)
net = resnet_v1_101_flownet_rfcn(cfg)
DT = torch.bfloat16 if os.environ.get('LSFA_PROBE_DTYPE') == 'bf16' else torch.float32      # bf16: the one-product mode (BASELINE configs[2])
key = net.get_key_test_symbol(cfg).bind(arg, aux, dev, dtype=DT)


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


for n in ([int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1, 2, 3, 4, 5, 6)):
    data = torch.rand(n, 3, H, W, device=dev) * 255
    data2 = torch.rand(n, 3, H, W, device=dev) * 255
    tb = graph_time(lambda: key._backbone(data))
    tf = graph_time(lambda: key._flownet(data, data2))
    print('fronts per pass %d: backbone %8.1f us (%7.1f per frame)   flownet %7.1f us (%6.1f per frame)' % (n, tb, tb / n, tf, tf / n), flush=True)
key.check_status()
