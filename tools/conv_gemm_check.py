#!/usr/bin/env python
"""3x3 convolutions with few output pixels and many channels: library convolution (NCHW, channels_last)
vs im2col + tuned GEMM."""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lsfa_amd import tuning
tuning.enable()
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

for name, nb, cin, cout, h, w, dil in [('fuse_reduce_add', 1, 256, 1024, 38, 63, 1), ('Nq_conv1', 2, 1024, 256, 38, 63, 1),
                                        ('feat_conv_3x3', 1, 2048, 1024, 38, 63, 6), ('res4 conv2', 1, 256, 256, 38, 63, 1),
                                        ('res5 conv2', 1, 512, 512, 38, 63, 2), ('res3 conv2', 1, 128, 128, 75, 125, 1),
                                        ('res2 conv2', 1, 64, 64, 150, 250, 1)]:
    x = torch.randn(nb, cin, h, w, device=dev); wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.01
    xl, wl = x.contiguous(memory_format=torch.channels_last), wt.contiguous(memory_format=torch.channels_last)
    w2 = wt.view(cout, -1)
    a = t(lambda: F.conv2d(x, wt, None, 1, dil, dil))
    b = t(lambda: F.conv2d(xl, wl, None, 1, dil, dil))
    def gemm():
        col = F.unfold(x, 3, dilation=dil, padding=dil)          # (nb, cin*9, L)
        return torch.matmul(w2, col)
    c = t(gemm)
    cu = t(lambda: F.unfold(x, 3, dilation=dil, padding=dil))
    # rows form: col_t (L, cin*9) x (cin*9, cout)
    colt = F.unfold(x, 3, dilation=dil, padding=dil)[0].t().contiguous(); w2t = w2.t().contiguous()
    d = t(lambda: torch.mm(colt, w2t))
    gf = 2 * 9 * cin * cout * h * w * nb / 1e9
    print('%-16s %5.1f GFLOP: NCHW %7.1f  NHWC %7.1f  unfold+GEMM %7.1f (unfold %5.1f)  row GEMM alone %7.1f us (%.0f TFLOP/s)' %
          (name, gf, a, b, c, cu, d, gf / d * 1e3 / 1e3 * 1e3 / 1e3), flush=True)
