#!/usr/bin/env python
"""Diagnostic: phase cycle shares of the proposal kernel on the bench's own frames."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lsfa_amd import hip
class A: pass
a = A(); a.interval=10; a.height=600; a.width=1000; a.dtype='f32'; a.no_graph=True; a.no_prefetch=True; a.steps=1; a.warmup=0; a.max_unique_steps=1
r = bench.Runner(a, 0, 'cuda:0')
r.prime()
stamps = torch.zeros(16, dtype=torch.int64, device='cuda:0')
hip.lib().lsfa_debug_set_proposal_stamps(ctypes.c_void_p(stamps.data_ptr()))
names = ['load keys', 'radix select', 'compaction', 'sort', 'nms', 'output']
for f in (1, 2, 5):
    if f == 1:
        r.fg.key_frame(r.frames[1])
    else:
        r.fg.cur_frame(r.frames[f], r.mv[f], r.res[f])
    torch.cuda.synchronize()
    s = stamps.cpu().numpy()
    d = np.diff(s[:7])
    print('frame', f, 'survivors', int(s[7]), 'total', int(s[6]-s[0]), {n: int(x) for n, x in zip(names, d)}, 'blocks', int(s[8]), 'A', int(s[9]), 'B', int(s[10]), 'C', int(s[11]))
