python - <<'PY'
import torch, sys
sys.path.insert(0,'.')
from lsfa_amd import hip
dev='cuda:0'
g=torch.Generator(device=dev).manual_seed(0)
def t(fn,n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
shapes=[('feat 3x3 d6 2048->1024',(38,63,2048,1024,3,1,6,6)),('fuse 3x3 256->1024',(38,63,256,1024,3,1,1,1)),('dcn gemm 4608->512',(38,63,4608,512,1,1,0,1)),('res4 conv2 256->256',(38,63,256,256,3,1,1,1)),('res5 conv1 2048->512',(38,63,2048,512,1,1,0,1))]
for name,(H,W,ci,co,k,st,pad,dil) in shapes:
    x=torch.randn((1,H,W,ci),device=dev,generator=g)
    w=hip.SplitWeight(torch.randn((co,ci,k,k),device=dev,generator=g)*0.01)
    print('%-26s %8.1f us'%(name,t(lambda: hip.conv_split(x,w,None,st,pad,dil,relu=True))),flush=True)
PY
