#!/usr/bin/env python
"""Analyse the wrong deform_im2col_cl outputs kept by LSFA_DCN_CHECK=1 (gpurun_out/garbage/*.npz): where do col and col_again differ,
which of them is right (recomputed now, alone on the GPU), and what do the wrong values look like?"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lsfa_amd import hip      # noqa: E402

for f in sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', 'garbage', '*.npz')))[:12]:
    d = np.load(f)
    c1, off = torch.from_numpy(d['c1']).cuda(), torch.from_numpy(d['off']).cuda()
    dil = 2 if ('u4_02' in f or 'u4_03' in f) else 1
    truth = hip.deform_im2col_cl(c1, off, 3, 3, dil, 1, dil, 4).cpu().numpy()
    C = c1.shape[3]
    print(os.path.basename(f), 'c1', tuple(c1.shape), 'off nonzero:', int((d['off'] != 0).sum()))
    for name in ('col', 'col2'):
        x = d[name]
        bad = np.argwhere(x != truth)
        if len(bad) == 0:
            print('   %s: correct' % name)
            continue
        rows = np.unique(bad[:, 1])
        k = bad[:, 2]
        taps, chans = np.unique(k // C), np.unique(k % C)
        wrong = x[x != truth]
        print('   %s: %d wrong elements in %d pixel rows (first %s last %s), taps %s, channels %d..%d (%d distinct); wrong values: %d zeros, |max| %.3g; truth there |max| %.3g' % (
            name, len(bad), len(rows), rows[:6].tolist(), rows[-3:].tolist(), taps.tolist(), chans.min(), chans.max(), len(chans),
            int((wrong == 0).sum()), float(np.abs(wrong).max()), float(np.abs(truth[x != truth]).max())))
        # are the wrong values the truth of another position?  compare with truth shifted by whole rows
        r0 = rows[0]
        seg = x[0, r0]
        match = [int(r) for r in range(max(0, r0 - 64), min(truth.shape[1], r0 + 64)) if np.array_equal(truth[0, r], seg)]
        print('      first wrong row %d equals truth row(s): %s' % (r0, match))
        # contiguous runs in flat index space
        flat = np.flatnonzero((x != truth).ravel())
        runs = np.split(flat, np.where(np.diff(flat) > 1)[0] + 1)
        print('      %d contiguous runs; lengths (floats) %s; starts %% 64 floats: %s' % (len(runs), [len(r) for r in runs[:8]], [int(r[0] % 64) for r in runs[:8]]))
