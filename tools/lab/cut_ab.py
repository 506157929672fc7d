#!/usr/bin/env python
"""A/B of two builds of the library on the same convolutions: prints a digest of every output (bit-identity across builds is read off
the digests) and the time per call.  LSFA_HIP_LIBRARY selects the build.
    python3 tools/lab/cut_ab.py; LSFA_HIP_LIBRARY=$PWD/tools/lab/_build/mix/liblsfa_hip_mix.so python3 tools/lab/cut_ab.py
"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lsfa_amd import hip  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(0)
CASES = [  # name, N, H, W, ci, co, k, dil
    ("feat x6", 6, 38, 63, 2048, 1024, 3, 6),
    ("res4 conv2 x6", 6, 38, 63, 256, 256, 3, 1),
    ("res4 conv1 x6", 6, 38, 63, 1024, 256, 1, 1),
    ("res4 conv3 x6", 6, 38, 63, 256, 1024, 1, 1),
    ("res3 conv2 x6", 6, 75, 125, 128, 128, 3, 1),
    ("res2 conv3 x6", 6, 150, 250, 64, 256, 1, 1),
    ("small feat x9", 9, 38, 63, 256, 1024, 3, 1),
    ("rfcn x9", 9, 38, 63, 1024, 1920, 1, 1),
    ("tiny values", 1, 38, 63, 256, 256, 3, 1),
]
if os.environ.get('CUT_AB_NT2'):      # the launches the plan runs on 128 x 64 tiles
    CASES = [
        ("res2 conv2 x6", 6, 150, 250, 64, 64, 3, 1),
        ("res2 conv1 x6", 6, 150, 250, 256, 64, 1, 1),
        ("res3 conv1 x6", 6, 75, 125, 512, 128, 1, 1),
        ("res3 conv2 x6", 6, 75, 125, 128, 128, 3, 1),
        ("res3 conv3 x6", 6, 75, 125, 128, 512, 1, 1),
        ("small conv2 x9", 9, 38, 63, 64, 64, 3, 1),
        ("small conv3 x9", 9, 38, 63, 64, 256, 1, 1),
        ("res4 conv2 x1", 1, 38, 63, 256, 256, 3, 1),
        ("res5 conv1 x1", 1, 38, 63, 2048, 512, 1, 1),
    ]
print("library:", hip.LIB_PATH)
for name, N, H, W, ci, co, k, dil in CASES:
    x = torch.relu(torch.randn((N, H, W, ci), device=DEV, generator=g))
    if name == "tiny values":
        x = x * torch.exp2(torch.randint(-40, 3, x.shape, device=DEV, generator=g).float())      # lo pieces across the subnormal range
    w = torch.randn((co, ci, k, k), device=DEV, generator=g) * 0.02
    b = torch.randn(co, device=DEV, generator=g)
    sw = hip.SplitWeight(w, pieces=2)
    am = hip.amax_partial(x)
    line = "%-16s" % name
    plans = [tuple(int(v) for v in q.split(',')) for q in os.environ['CUT_AB_PLANS'].split(';')] if os.environ.get('CUT_AB_PLANS') else [(2, 4, 3, 1), (2, 4, 4, 1), (1, 4, 2, 1)]
    for plan in [None] + plans:      # the plan's own choice; then forced (kernel, nt, st, slices): 2 = loader / consumer waves, 1 = mixed roles
        if plan is None:
            hip.conv_plan_override()
        else:
            hip.conv_plan_override(kernel=plan[0], nt=plan[1], st=plan[2], slices=plan[3])
        y = hip.conv_split(x, sw, b, 1, dil * (k // 2), dil, relu=True, amax_in=am)
        torch.cuda.synchronize()
        digest = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:8]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            hip.conv_split(x, sw, b, 1, dil * (k // 2), dil, relu=True, amax_in=am)
        e0.record()
        for _ in range(20):
            hip.conv_split(x, sw, b, 1, dil * (k // 2), dil, relu=True, amax_in=am)
        e1.record()
        torch.cuda.synchronize()
        line += "  | %s %s %7.1f us" % ("plan" if plan is None else "%d,%d,%d,%d" % plan, digest, e0.elapsed_time(e1) * 1000 / 20)
    hip.conv_plan_override()
    print(line)
