#!/usr/bin/env python
"""Which frames differ between `python -m lsfa_amd.test` runs that should agree?  

Runs the command of tests/test_multirank_gpu.py as child processes under different conditions — cold / warm MIOpen
caches (HOME redirected), one rank, two ranks under torch.distributed.run, two independent single-rank processes at
the same time, forced stream layouts, serial loop — and prints, per pair of runs, the number of detection rows
without a bit-identical counterpart, per global frame id.  Output: gpurun_out/diag/*.npy + a table on stdout.
"""
import os
import shutil
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, 'gpurun_out', 'diag')
HOMES = '/tmp/lsfa_diag_homes'      # MIOpen caches: large, not worth copying back
ARGS = ['--clips', '3', '--frames', '7', '--interval', '3', '--height', '192', '--width', '320']


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def env_for(home, extra=None):
    env = dict(os.environ)
    env.update(LSFA_BENCH_BACKEND='gloo', LSFA_BENCH_ONE_DEVICE='1', MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY='0',
               LSFA_LOG_LAYOUT='1', PYTHONPATH=ROOT + os.pathsep + env.get('PYTHONPATH', ''))
    if home:
        os.makedirs(home, exist_ok=True)
        env['HOME'] = home
    env.update(extra or {})
    return env


def launch(tag, nproc, home, extra=None, more=()):
    out = os.path.join(OUT, 'rows_%s.npy' % tag)
    if nproc == 1:
        cmd = [sys.executable, '-m', 'lsfa_amd.test'] + ARGS + list(more) + ['--out', out]
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc), '--master-addr',
               '127.0.0.1', '--master-port', str(free_port()), '-m', 'lsfa_amd.test'] + ARGS + list(more) + ['--out', out]
    log = open(os.path.join(OUT, 'log_%s.txt' % tag), 'w')
    return subprocess.Popen(cmd, cwd=ROOT, env=env_for(home, extra), stdout=log, stderr=subprocess.STDOUT), out, tag


def wait(procs):
    res = {}
    for p, out, tag in procs:
        t0 = time.time()
        rc = p.wait(timeout=1500)
        lay = [l.strip() for l in open(os.path.join(OUT, 'log_%s.txt' % tag)) if 'streams:' in l]
        print('run %-12s rc %d  (%.0f s)  %s' % (tag, rc, time.time() - t0, ' | '.join(lay)), flush=True)
        res[tag] = np.load(out) if rc == 0 and os.path.exists(out) else None
    return res


def per_frame_diff(a, b):
    """rows of a (per frame) that have no bit-identical row in b"""
    sb = {tuple(r) for r in b}
    bad = {}
    for r in a:
        if tuple(r) not in sb:
            bad[int(r[0])] = bad.get(int(r[0]), 0) + 1
    return bad


def main():
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    runs = {}
    shutil.rmtree(HOMES, ignore_errors=True)
    h1, h2, h3, h4 = (os.path.join(HOMES, 'home%d' % i) for i in (1, 2, 3, 4))
    shared = {}
    runs.update(wait([launch('one_cold', 1, h1)]))
    runs.update(wait([launch('sh_one_cold', 1, h4, shared)]))
    runs.update(wait([launch('sh_two_cold', 2, h2, shared)]))
    runs.update(wait([launch('sh_one_after', 1, h2, shared)]))
    runs.update(wait([launch('two_warm', 2, h1)]))
    runs.update(wait([launch('one_warm', 1, h1)]))
    runs.update(wait([launch('two_cold', 2, h2)]))
    runs.update(wait([launch('pair_a', 1, h3), launch('pair_b', 1, h3)]))           # two independent processes, cold, at once
    runs.update(wait([launch('serial', 1, h1, more=['--serial'])]))
    runs.update(wait([launch('plain', 1, h1, {'LSFA_STREAM_LAYOUT': 'plain'})]))
    runs.update(wait([launch('onequeue', 1, h1, {'LSFA_STREAM_LAYOUT': 'one-queue'})]))
    runs.update(wait([launch('two_onequeue', 2, h1, {'LSFA_STREAM_LAYOUT': 'one-queue'})]))
    runs.update(wait([launch('two_plain', 2, h1, {'LSFA_STREAM_LAYOUT': 'plain'})]))
    ref = runs['one_cold']
    print('\nrows per run:', {k: (None if v is None else len(v)) for k, v in runs.items()})
    for k, v in runs.items():
        if v is None or k == 'one_cold':
            continue
        d1, d2 = per_frame_diff(ref, v), per_frame_diff(v, ref)
        print('one_cold vs %-12s: %5d / %5d rows not bit-identical; per frame (ref->run): %s' % (
            k, sum(d1.values()), sum(d2.values()), dict(sorted(d1.items()))))
        if d1:
            # size of the differences: nearest row of the same frame and class
            worst = 0.0
            for r in ref:
                if int(r[0]) in d1:
                    cand = v[(v[:, 0] == r[0]) & (v[:, 1] == r[1])]
                    if len(cand):
                        worst = max(worst, float(np.abs(cand[:, 2:] - r[2:]).max(1).min()))
            print('    largest nearest-row distance (score/px): %.3g' % worst)


if __name__ == '__main__':
    main()
