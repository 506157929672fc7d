#!/usr/bin/env python
"""lsfa_stem_conv7x7s2 at the batch shapes of the pipeline (key bank, FlowNet pair, small net of a segment), hipGraph-timed."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lsfa_amd import hip
if os.environ.get('STEM_LIB'):
    hip.LIB_PATH = os.environ['STEM_LIB']      # an ablation build of the library (tools/lab/_build)
dev = 'cuda:0'
w = hip.stem_weight_layout(torch.randn(64, 3, 7, 7, device=dev) * 0.05)
b = torch.randn(64, device=dev)
sc, sh = torch.full((3,), 0.02, device=dev), torch.full((3,), -2.0, device=dev)
for N, H, W in ((1, 600, 1000), (6, 600, 1000), (6, 300, 500), (9, 150, 250)):
    x = torch.rand(N, 3, H, W, device=dev) * 255
    out = hip.stem_conv(x, w, b, sc, sh)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hip.stem_conv(x, w, b, sc, sh, out=out)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(10):
                hip.stem_conv(x, w, b, sc, sh, out=out)
        g.replay(); s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s); s.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    fl = 2.0 * 147 * 64 * N * out.shape[1] * out.shape[2]
    print('stem %d x %dx%d: %.1f us (%.1f per image)  %.1f TFLOP/s   output %.0f GB/s' % (N, H, W, us, us / N, fl / us / 1e6, out.numel() * 4 / us / 1e3))
