#!/usr/bin/env python
"""Is the host ahead of the GPU in the pipelined bench loop?  Queues `--steps` intervals through bench.Runner exactly like bench.py's timed
region and prints when (ms after the region start) the host entered each step, next to when the GPU finished the region."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench  # noqa: E402

sys.argv = ['bench.py', '--steps', '24', '--no-cpu-baseline', '--no-parity'] + sys.argv[1:]
args = bench.parse()
r = bench.Runner(args, 0, 'cuda:0')
r.prime()


def drain():
    r.fg.flush()
    torch.cuda.synchronize()


for s in range(32):
    r.step(s, end=32)
drain()
for rep in range(2):
    t0 = time.perf_counter()
    marks = []
    for s in range(args.steps):
        marks.append((time.perf_counter() - t0) * 1e3)
        r.step(s, end=args.steps)
    t_host = (time.perf_counter() - t0) * 1e3
    drain()
    t_all = (time.perf_counter() - t0) * 1e3
    print('host entered the steps at (ms): ' + ' '.join('%.1f' % m for m in marks))
    print('host done queueing at %.1f ms, GPU done at %.1f ms (%.0f frames/s)' % (t_host, t_all, args.steps * args.interval / t_all * 1e3), flush=True)
