#!/usr/bin/env python
"""How long the host spends queueing one step (1 key + 9 non-key frames) vs how long the GPU takes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
sys.argv = ['bench.py', '--no-cpu-baseline'] + sys.argv[1:]

def drain():
    for x in [r]:
        getattr(x.fg, 'flush', lambda: None)()     # the pipeline may hold a segment back (look-ahead mode)
    torch.cuda.synchronize()

args = bench.parse()
r = bench.Runner(args, 0, 'cuda:0')
r.prime()
for s in range(3):
    r.step(s)
drain()
# (a) queue one step at a time and wait: host time to queue, then GPU completion
tq = tg = 0.0
for s in range(3, 13):
    t0 = time.perf_counter(); r.step(s); t1 = time.perf_counter(); drain(); t2 = time.perf_counter()
    tq += t1 - t0; tg += t2 - t0
print('one step at a time: host queues a step in %.2f ms; step done after %.2f ms' % (tq / 10 * 1e3, tg / 10 * 1e3))
# (b) free running
t0 = time.perf_counter()
for s in range(13, 33):
    r.step(s)
t1 = time.perf_counter(); drain(); t2 = time.perf_counter()
print('20 steps free running: host loop %.2f ms/step, wall %.2f ms/step' % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
