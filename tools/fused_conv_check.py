#!/usr/bin/env python
"""3x3 convolution + separate bias/ReLU pass vs torch.miopen_convolution_relu (channels-last, hipGraph replay):
the fused library op is a naive kernel here (4-22 ms against 44-48 us), which is why the pass is hand-written."""
import torch, torch.nn.functional as F, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lsfa_amd import hip
torch.backends.cudnn.benchmark = True
dev='cuda:0'
def t(fn, n=40):
    for _ in range(6): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(10): fn()
    g.replay(); torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n/10*1e3
cl=torch.channels_last
for cin,cout,h,w,dil in [(256,256,38,63,1),(128,128,75,125,1),(64,64,150,250,1)]:
    x=torch.randn(1,cin,h,w,device=dev).contiguous(memory_format=cl); wt=(torch.randn(cout,cin,3,3,device=dev)*0.01).contiguous(memory_format=cl)
    b=torch.randn(cout,device=dev); ones=torch.ones(cout,device=dev)
    def sep():
        y=F.conv2d(x,wt,None,1,dil,dil)
        r=y.permute(0,2,3,1).reshape(-1,cout)
        hip.scale_shift_relu_cl(r,ones,b,True,out=r)
        return y
    def fused():
        return torch.miopen_convolution_relu(x,wt,b,[1,1],[dil,dil],[dil,dil],1)
    ts=t(sep)
    try:
        tf=t(fused)
        err=(sep()-fused()).abs().max().item()
    except Exception as e:
        tf=float('nan'); err=str(e)[:80]
    print('%d->%d %dx%d: conv + bias/relu pass %.1f us; miopen_convolution_relu %.1f us; max diff %s'%(cin,cout,h,w,ts,tf,err),flush=True)
