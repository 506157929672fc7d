#!/usr/bin/env python
"""One shape, a few forced plans, each run `reps` times eagerly with a device synchronise in between: for `rocprofv3 --kernel-trace`
(per-kernel durations of the main kernel and its reduce pass, without launch gaps or host effects).
    rocprofv3 --kernel-trace --stats -d gpurun_out/trace -- python3 tools/lab/conv_trace_probe.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lsfa_amd import hip  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
KERN = int(os.environ.get('PROBE_KERNEL', '1'))      # 1: mixed-role waves, 2: loader / consumer waves
CASES = [  # name, H, W, ci, co, k, dil, epi, plans (nt, st, s)
    ("res4 conv1", 38, 63, 1024, 256, 1, 1, 'relu', [(2, 3, 3), (2, 3, 1), (4, 3, 6)]),
    ("res4 conv2", 38, 63, 256, 256, 3, 1, 'relu', [(2, 3, 3), (4, 2, 12)]),
    ("res4 conv3", 38, 63, 256, 1024, 1, 1, 'res2', [(2, 3, 1)]),
    ("res4 conv3 plain", 38, 63, 256, 1024, 1, 1, 'none', [(2, 3, 1)]),
    ("small 1x1 64->256", 38, 63, 64, 256, 1, 1, 'res2', [(2, 2, 1)]),
    ("feat", 38, 63, 2048, 1024, 3, 6, 'relu', [(4, 3, 3)]),
]
for name, H, W, ci, co, k, dil, epi, plans in CASES:
    x = torch.relu(torch.randn((1, H, W, ci), device=DEV, generator=g))
    w = torch.randn((co, ci, k, k), device=DEV, generator=g) * 0.02
    b = torch.randn(co, device=DEV, generator=g)
    res = torch.randn((1, H, W, co), device=DEV, generator=g)
    sc2, sh2 = torch.rand(co, device=DEV, generator=g) + 0.5, torch.randn(co, device=DEV, generator=g)
    sw = hip.SplitWeight(w, pieces=2)
    am = hip.amax_partial(x)
    slots = hip.amax_slots(1, DEV)[0]
    for (nt, st, s) in plans:
        hip.conv_plan_override(kernel=KERN, nt=nt, st=st, slices=s)
        marker = torch.zeros(nt * 100 + st * 10 + s, device=DEV)       # a fill kernel whose size tags the plan in the trace
        for _ in range(6):
            kw = dict(amax_in=am, amax_out=slots)
            if epi == 'res2':
                kw.update(residual=res, out2=torch.empty_like(res), scale2=sc2, shift2=sh2)
            hip.conv_split(x, sw, b, 1, dil * (k // 2), dil, relu=epi == 'relu', **kw)
            torch.cuda.synchronize()
    hip.conv_plan_override()
print("done")
