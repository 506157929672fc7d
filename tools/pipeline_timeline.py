#!/usr/bin/env python
"""Durations of the pipeline's graph replays while the whole pipeline is running (HIP events on
each replay's own stream): where does a step's time go under contention?"""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from lsfa_amd import tuning
from lsfa_amd.core import graphs
tuning.enable()
torch.backends.cudnn.benchmark = True
sys.argv = ['bench.py', '--no-cpu-baseline'] + sys.argv[1:]
args = bench.parse()
r = bench.Runner(args, 0, 'cuda:0')
r.prime()
for s in range(3):
    r.step(s)
r.fg.flush()
torch.cuda.synchronize()
rec = collections.defaultdict(list)
t_origin = torch.cuda.Event(enable_timing=True)

def timed(name, fn):
    def wrapper(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(*a, **k); e1.record()
        rec[name].append((e0, e1))
        return out
    return wrapper

fp = r.fg
for lane in fp.klanes:
    for n in ('run_front', 'run_flow', 'run_agg', 'run_tail'):
        setattr(lane, n, timed(n[4:], getattr(lane, n)))
for i, lane in enumerate(fp.lanes):
    lane.cur_frame = timed('cur_frame', lane.cur_frame)
t_origin.record()
n = 10
t0 = time.perf_counter()
for s in range(3, 3 + n):
    r.step(s)
fp.flush()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / n * 1e3
print('wall %.2f ms/step' % wall)
for name, evs in rec.items():
    d = [a.elapsed_time(b) for a, b in evs]
    st = [t_origin.elapsed_time(a) for a, _ in evs]
    print('%-10s n=%3d  mean %.3f ms  min %.3f  max %.3f   per step %.2f ms' % (name, len(d), sum(d) / len(d), min(d), max(d), sum(d) / n))
# timeline of one steady-state step (the 6th): start/end relative to its front start
k = 5
f0 = rec['front'][k][0]
for name in ('front', 'flow', 'agg', 'tail'):
    a, b = rec[name][k]
    print('step %d %-6s start %7.3f end %7.3f' % (k, name, f0.elapsed_time(a), f0.elapsed_time(b)))
a, b = rec['front'][k + 1]
print('step %d front  start %7.3f end %7.3f' % (k + 1, f0.elapsed_time(a), f0.elapsed_time(b)))
print('non-key frames of step %d (queued after key frame %d):' % (k, k + 1))
for j in range(9 * k, 9 * k + 9):
    a, b = rec['cur_frame'][j]
    print('   cur %d start %7.3f end %7.3f' % (j - 9 * k + 1, f0.elapsed_time(a), f0.elapsed_time(b)))
