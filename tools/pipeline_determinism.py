#!/usr/bin/env python
"""Run the same 11 frames twice through the serial graphs and through FramePipeline; report
per-frame mismatches (a race between streams shows up as run-to-run differences)."""
import sys
import numpy as np, torch
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lsfa_amd.config.config import lsfa_test_config
from lsfa_amd.core.graphs import FrameGraphs, FramePipeline
from lsfa_amd.symbols import params as P
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
from lsfa_amd.utils.synthetic import SyntheticClip
DEV = 'cuda:0'
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (192, 320)
cfg = lsfa_test_config(key_frame_interval=10)
arg, aux = P.init_params(cfg, seed=3)
net = resnet_v1_101_flownet_rfcn(cfg)
key = net.get_key_test_symbol(cfg).bind(arg, aux, DEV)
cur = net.get_cur_test_symbol(cfg).bind(arg, aux, DEV)
clip = SyntheticClip(0, 12, H, W)
sched = [(f, 1 + 4 * ((f - 1) // 4)) for f in range(1, 12)]
frames = {f: clip.frame(f, DEV) for f in range(12)}
mvs = {f: clip.motion_vector(f, kf, DEV) for f, kf in sched if f != kf}
ress = {f: clip.res_diff(f, DEV) for f, kf in sched if f != kf}
torch.cuda.synchronize()

def run(fp, pipelined):
    outs = {}
    def keep(f):
        def deliver(bufs):
            outs[f] = (bufs[0].clone(), bufs[1].clone())
        return deliver
    fp.first_frame(frames[0])
    if not (fp.captured if pipelined else fp.key_graph is not None):
        fp.capture()
    for f, kf in sched:
        if pipelined:
            if f == kf: fp.key_frame(frames[f], deliver=keep(f))
            else: fp.cur_frame(frames[f], mvs[f], ress[f], deliver=keep(f))
        else:
            b = fp.key_frame(frames[f]) if f == kf else fp.cur_frame(frames[f], mvs[f], ress[f])
            keep(f)(b)
    if pipelined: fp.join()
    torch.cuda.synchronize()
    return {f: (d.cpu().numpy(), c.cpu().numpy()) for f, (d, c) in outs.items()}

def compare(name, a, b):
    bad = []
    for f, kf in sched:
        n = int((a[f][0] != b[f][0]).sum())
        if n or (a[f][1] != b[f][1]).any():
            bad.append((f, 'key' if f == kf else 'cur', n))
    print(name, 'mismatching frames:', bad if bad else 'none', flush=True)

fg = FrameGraphs(key, cur, cfg, H, W, DEV, prefetch=False)
serial = run(fg, False)
compare('serial graphs, run 1 vs 2', serial, run(fg, False))
for lanes in (1, 3):
    fp = FramePipeline(key, cur, cfg, H, W, DEV, lanes=lanes)
    r1, r2, r3 = run(fp, True), run(fp, True), run(fp, True)
    compare('pipeline lanes=%d, run 1 vs 2' % lanes, r1, r2)
    compare('pipeline lanes=%d, run 2 vs 3' % lanes, r2, r3)
    compare('pipeline lanes=%d vs serial' % lanes, r3, serial)
