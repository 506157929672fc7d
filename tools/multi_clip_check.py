#!/usr/bin/env python
"""Aggregate frames/s of K independent clips pipelined on ONE GPU (is a single clip's pipeline limited
by its dependency chain or by the GPU?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
K = int(sys.argv[1]); rest = sys.argv[2:]
sys.argv = ['bench.py', '--no-cpu-baseline', '--max-unique-steps', '4'] + rest

def drain():
    for x in runners:
        getattr(x.fg, 'flush', lambda: None)()     # the pipeline may hold a segment back (look-ahead mode)
    torch.cuda.synchronize()

args = bench.parse()
runners = []
for k in range(K):
    r = bench.Runner(args, k, 'cuda:0'); r.prime(); runners.append(r)
for s in range(3):
    for r in runners: r.step(s)
drain()
n = 20
t0 = time.perf_counter()
for s in range(3, 3 + n):
    for r in runners: r.step(s)
drain()
dt = time.perf_counter() - t0
print('%d clips: %.1f frames/s aggregate, %.2f ms per (step of every clip)' % (K, K * n * args.interval / dt, dt / n * 1e3))
