#!/usr/bin/env python
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lsfa_amd.core import streams
dev = torch.device('cuda:0')
pool = [torch.cuda.Stream(device=dev) for _ in range(10)]
t0 = time.time()
print('overlap matrix of 10 consecutive pool streams (1 = concurrent):')
for i, a in enumerate(pool):
    print(' '.join('1' if (i != j and streams.overlaps(a, b, dev)) else ('.' if i == j else '0') for j, b in enumerate(pool)))
print('probe time %.2f s' % (time.time() - t0))
t0 = time.time()
chosen, aliased = streams.concurrent_streams(6, dev)
print('chosen', len(chosen), 'aliased', [a for _, a in aliased], 'in %.2f s' % (time.time() - t0))
print('default stream vs chosen:', [streams.overlaps(torch.cuda.default_stream(dev), c, dev) for c in chosen])
