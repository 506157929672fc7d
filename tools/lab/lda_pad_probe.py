#!/usr/bin/env python
"""Lab: does the pixel stride of a channels-last activation map matter?  With Cin a power of two (1024 channels = 4096 bytes per pixel)
every pixel's 128-byte piece of one K chunk sits at the same offset modulo 4 KB: if the L2's channels interleave at that granularity, all the
A-tile requests of a chunk queue on one channel.  The same convolution is timed with the input map padded to Cin + pad floats per pixel
(lda of the descriptor; the padding is never read)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lsfa_amd import hip  # noqa: E402
from conv_ring_lab import timed  # noqa: E402

DEV = "cuda:0"
SHAPES = {
    "res4 conv1 1024->256": (38, 63, 1024, 256, 1, 1),
    "res4 conv2 256->256 3x3": (38, 63, 256, 256, 3, 1),
    "res4 conv3 256->1024": (38, 63, 256, 1024, 1, 1),
    "res5 conv1 2048->512": (38, 63, 2048, 512, 1, 1),
    "res5 conv3 512->2048": (38, 63, 512, 2048, 1, 1),
    "res3 conv1 512->128": (75, 125, 512, 128, 1, 1),
    "feat 3x3 d6 2048->1024": (38, 63, 2048, 1024, 3, 6),
}
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 3
g = torch.Generator(device=DEV).manual_seed(0)
for name, (H, W, ci, co, k, dil) in SHAPES.items():
    pad = dil * (k // 2)
    ws = [torch.randn((co, ci, k, k), device=DEV, generator=g) * (1.0 / (ci * k * k) ** 0.5) for _ in range(4)]
    sws = [hip.SplitWeight(w, pieces=2) for w in ws]
    b = torch.randn(co, device=DEV, generator=g)
    row = []
    for extra in (0, 32, 64, 96):
        xs = [torch.zeros((NB, H, W, ci + extra), device=DEV) for _ in range(3)]
        for x in xs:
            x[..., :ci] = torch.relu(torch.randn((NB, H, W, ci), device=DEV, generator=g)) * 2.0
        ams = [hip.amax_partial(x) for x in xs]
        out = torch.empty((NB, H, W, co), device=DEV)

        def call(i):
            return hip.conv_split_view(xs[i % 3], sws[i % 4], b, out, stride=1, pad=(pad, pad), dil=dil, act=1, cin=ci, amax_in=ams[i % 3])
        call(0)
        row.append((extra, timed(call, 16)))
    print("%-28s %d image(s): " % (name, NB) + "   ".join("pad %3d: %7.1f us" % r for r in row), flush=True)
