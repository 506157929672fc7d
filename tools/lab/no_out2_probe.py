#!/usr/bin/env python
"""Upper bound of what not materialising relu(bn1_next(sum)) (conv3's second output) could save: the backbone at G images per pass,
hipGraph-timed, as built vs with conv3 writing only the sum and the next conv1 reading it (WRONG numbers: a timing probe only)."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lsfa_amd.config.config import lsfa_test_config
from lsfa_amd.symbols import params as P
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
dev = 'cuda:0'
cfg = lsfa_test_config(key_frame_interval=10)
arg, aux = P.init_params(cfg, seed=0)
key = resnet_v1_101_flownet_rfcn(cfg).get_key_test_symbol(cfg).bind(arg, aux, dev)


def timed(G, reps=10):
    data = torch.rand(G, 3, 600, 1000, device=dev) * 255
    s = torch.cuda.Stream()
    with torch.no_grad(), torch.cuda.stream(s):
        key._backbone(data)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            key._backbone(data)
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s); s.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


orig = key._conv


def no_out2(x, sw, bias=None, *a, **kw):
    if 'out2' in kw:
        kw.pop('out2'); kw.pop('scale2'); kw.pop('shift2')
        y = orig(x, sw, bias, *a, **kw)
        return y, y
    return orig(x, sw, bias, *a, **kw)


for G in (1, 6):
    t0 = timed(G)
    key._conv = no_out2
    t1 = timed(G)
    key._conv = orig
    print('backbone, %d images per pass: as built %.0f us (%.0f per image), conv3 without its second output %.0f us (%.0f per image)' % (G, t0, t0 / G, t1, t1 / G))
