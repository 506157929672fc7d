#!/usr/bin/env python
"""Does replaying independent frame graphs on two HIP streams raise throughput?
Non-key frames depend only on the key feature, so frame t's single-workgroup tail (Proposal,
head, detection NMS) can run beside frame t+1's convolutions."""
import sys, time
import torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lsfa_amd.config.config import lsfa_test_config
from lsfa_amd.core.graphs import FrameGraphs
from lsfa_amd.symbols import params as P
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
from lsfa_amd.utils.synthetic import SyntheticClip

dev = 'cuda:0'
H, W = 600, 1000
cfg = lsfa_test_config(key_frame_interval=10)
arg, aux = P.init_params(cfg, seed=0)
net = resnet_v1_101_flownet_rfcn(cfg)
key = net.get_key_test_symbol(cfg).bind(arg, aux, dev)
cur = net.get_cur_test_symbol(cfg).bind(arg, aux, dev)
clip = SyntheticClip(0, 12, H, W)
frames = [clip.frame(i).to(dev) for i in range(4)]
mv, res = clip.motion_vector(1, 0).to(dev), clip.res_diff(1).to(dev)
nstreams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
fgs, streams = [], []
for k in range(nstreams):
    fg = FrameGraphs(key, cur, cfg, H, W, dev, prefetch=False)
    fg.first_frame(frames[0]); fg.capture()
    fg.data.copy_(frames[1]); fg.mv.copy_(mv); fg.res.copy_(res)
    fgs.append(fg); streams.append(torch.cuda.Stream(device=dev))
torch.cuda.synchronize()

def run(n_streams, frames_total, which='cur'):
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(frames_total):
        k = i % n_streams
        with torch.cuda.stream(streams[k]):
            (fgs[k].cur_graph if which == 'cur' else fgs[k].key_graph).replay()
    torch.cuda.synchronize()
    return (time.time() - t0) / frames_total * 1e3

for which in ('cur', 'key'):
    for ns in range(1, nstreams + 1):
        run(ns, 20, which)
        print(which, 'streams', ns, 'ms/frame %.3f' % run(ns, 200 if which == 'cur' else 40, which), flush=True)
# mixed: step pattern 1 key + 9 cur, all independent (upper bound for pipelining within a clip)
for ns in range(1, nstreams + 1):
    def step():
        for i in range(10):
            k = i % ns
            with torch.cuda.stream(streams[k]):
                (fgs[k].key_graph if i == 0 else fgs[k].cur_graph).replay()
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): step()
    torch.cuda.synchronize()
    print('mixed streams', ns, 'ms/step %.3f' % ((time.time() - t0) / 20 * 1e3), flush=True)
