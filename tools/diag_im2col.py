#!/usr/bin/env python
"""Is lsfa_deform_im2col_cl's output ever wrong when another process shares the GPU?  (tools/diag_garbage.py points at it.)
Loop: a producer kernel rewrites the data map (alternating between two different contents), the offsets are rewritten, the
im2col runs, and its output is compared with the golden result for that content.  A child process keeps the GPU busy."""
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lsfa_amd import hip          # noqa: E402


def hammer():
    a = torch.randn(4096, 4096, device='cuda')
    b = torch.randn(4096, 4096, device='cuda')
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        for _ in range(20):
            c = a @ b
            c = torch.relu_(c)
        torch.cuda.synchronize()


def main():
    if len(sys.argv) > 1 and sys.argv[1] == 'hammer':
        return hammer()
    secs = 40
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), 'hammer', str(secs + 20)]) if os.environ.get('NO_HAMMER') != '1' else None
    dev = 'cuda:0'
    torch.manual_seed(0)
    for (H, W, C, dil) in ((24, 40, 128, 1), (12, 20, 512, 2), (75, 125, 128, 1)):
        src = [torch.randn(1, H, W, C, device=dev), torch.randn(1, H, W, C, device=dev)]
        off_src = torch.zeros(1, H, W, 128, device=dev)
        gold = []
        for k in range(2):
            gold.append(hip.deform_im2col_cl(src[k], off_src, 3, 3, dil, 1, dil, 4).clone())
        torch.cuda.synchronize()
        assert not torch.equal(gold[0], gold[1])
        data = torch.empty_like(src[0])
        off = torch.empty_like(off_src)
        bad = 0
        iters = 0
        t0 = time.time()
        first = None
        while time.time() - t0 < secs / 3.0:
            for it in range(50):
                k = (iters + it) & 1
                data.copy_(src[k])                  # the producer of this iteration
                off.copy_(off_src)
                col = hip.deform_im2col_cl(data, off, 3, 3, dil, 1, dil, 4)
                ok = torch.equal(col, gold[k])
                if not ok:
                    bad += 1
                    if first is None:
                        d = (col != gold[k]).nonzero()
                        same_as_other = int((col[col != gold[k]] == gold[1 - k][col != gold[k]]).sum())
                        first = 'iteration %d: %d elements differ, %d of them equal the OTHER content; first index %s, last %s' % (
                            iters + it, len(d), same_as_other, d[0].tolist(), d[-1].tolist())
            iters += 50
        print('H %d W %d C %d: %d iterations, %d wrong outputs. %s' % (H, W, C, iters, bad, first or ''), flush=True)
    if child is not None:
        child.wait()


if __name__ == '__main__':
    main()
