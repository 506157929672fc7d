#!/usr/bin/env python
"""Where a key frame's time goes: each section captured as its own hipGraph and replayed alone."""
import sys
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lsfa_amd.config.config import lsfa_test_config
from lsfa_amd.symbols import params as P
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
import os
dev = 'cuda:0'
H, W = 600, 1000
cfg = lsfa_test_config(key_frame_interval=10)
arg, aux = P.init_params(cfg, seed=0)
net = resnet_v1_101_flownet_rfcn(cfg)
key = net.get_key_test_symbol(cfg).bind(arg, aux, dev)
cur = net.get_cur_test_symbol(cfg).bind(arg, aux, dev)
data = torch.rand(1, 3, H, W, device=dev) * 255
data2 = torch.rand(1, 3, H, W, device=dev) * 255
im_info = torch.tensor([[H, W, 1.0]], device=dev)
feat_old = torch.randn(1, 1024, 38, 63, device=dev)
mv = torch.randn(1, 2, 38, 63, device=dev); res = torch.randn(1, 3, 38, 63, device=dev)

def graph_time(fn, n=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        fn()
    g.replay(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

with torch.no_grad():
    conv_feat, flow, scale = key._key_front(data, data2)
secs = [
    ('backbone (ResNet-101 + DCN + feat conv)', lambda: key._backbone(data)),
    ('flownet', lambda: key._flownet(data, data2)),
    ('key back (warp + Nq + aggregate + heads)', lambda: key._key_back(conv_feat, flow, scale, feat_old, im_info)),
    ('heads only (rpn + proposal + rfcn + psroi)', lambda: key._heads(conv_feat, im_info)),
    ('whole key frame', lambda: key.forward(data=data, im_info=im_info, data_key_old=data2, feat_key_old=feat_old)),
    ('small net feature', lambda: cur.small_net_feature(data)),
    ('whole non-key frame', lambda: cur.forward(data=data, im_info=im_info, feat_key=feat_old, motion_vector=mv, res_diff=res)),
]
import os
for name, fn in secs:
    print('%-46s %8.1f us' % (name, graph_time(fn)), flush=True)
