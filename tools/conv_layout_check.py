#!/usr/bin/env python
"""Time the backbone's convolution shapes in NCHW vs channels_last (MIOpen find on) and the 1x1
GEMM formulations: addmm on [C,HW] + separate bias/ReLU pass vs torch._addmm_activation on [HW,C]."""
import sys, time
import torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = 'cuda:0'

def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

shapes3 = [(64, 64, 150, 250, 1, 1), (128, 128, 75, 125, 1, 1), (256, 256, 38, 63, 1, 1), (512, 512, 38, 63, 2, 1),
           (1024, 256, 38, 63, 1, 2), (2048, 1024, 38, 63, 6, 1), (3, 64, 600, 1000, 1, 1)]
for cin, cout, h, w, dil, nb in shapes3:
    k = 7 if cin == 3 else 3
    stride = 2 if cin == 3 else 1
    pad = 3 if cin == 3 else dil
    x = torch.randn(nb, cin, h, w, device=dev); wt = torch.randn(cout, cin, k, k, device=dev) * 0.01
    xl, wl = x.contiguous(memory_format=torch.channels_last), wt.contiguous(memory_format=torch.channels_last)
    a = t(lambda: F.conv2d(x, wt, None, stride, pad, dil))
    b = t(lambda: F.conv2d(xl, wl, None, stride, pad, dil))
    print('conv%dx%d %4d->%4d %3dx%3d dil %d n%d: NCHW %7.1f us   NHWC %7.1f us' % (k, k, cin, cout, h, w, dil, nb, a, b), flush=True)

for cin, cout, hw in [(64, 256, 150 * 250), (256, 64, 150 * 250), (256, 128, 75 * 125), (128, 512, 75 * 125), (512, 256, 38 * 63),
                      (256, 1024, 38 * 63), (1024, 256, 38 * 63), (1024, 512, 38 * 63), (512, 2048, 38 * 63), (2048, 512, 38 * 63)]:
    X = torch.randn(cin, hw, device=dev); Wm = torch.randn(cout, cin, device=dev) * 0.01; bias = torch.randn(cout, device=dev)
    Xt = X.t().contiguous(); Wt = Wm.t().contiguous()
    out = torch.empty(cout, hw, device=dev)
    def nchw():
        torch.mm(Wm, X, out=out)
        out.add_(bias[:, None]).relu_()
    a = t(nchw)
    a0 = t(lambda: torch.mm(Wm, X, out=out))
    b = t(lambda: torch._addmm_activation(bias, Xt, Wt))
    b0 = t(lambda: torch.mm(Xt, Wt))
    print('1x1 %4d->%4d hw %6d: [C,HW] mm %6.1f (+bias/relu pass %6.1f)   [HW,C] mm %6.1f  addmm_activation %6.1f us' % (cin, cout, hw, a0, a, b0, b), flush=True)
