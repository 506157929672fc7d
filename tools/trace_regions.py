"""per-interval timing of the three timed regions of bench.py's loop (key-stream events), to see where a slow first region loses its time"""
import os, sys, time, json
sys.argv = ['bench.py', '--gpus', '1', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-parity']
sys.path.insert(0, '.')
import torch
import bench
args = bench.parse()
r = bench.Runner(args, 0, 'cuda:0')
r.prime()
def drain():
    r.fg.flush(); torch.cuda.synchronize()
t = time.perf_counter(); n = 0
SETTLE = float(os.environ.get('SETTLE', '2.5'))
DR = int(os.environ.get('DR', '8'))
while time.perf_counter() - t < SETTLE:
    r.step(n); n += 1
    if DR and n % DR == 0: drain()
drain()
W = int(os.environ.get('W', '5'))
for s in range(W): r.step(s)
drain()
out = []
for rep in range(3):
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True); e0.record(r.fg.s_key)
    for s in range(5, 25):
        r.step(s)
        e = torch.cuda.Event(enable_timing=True); e.record(r.fg.s_key); evs.append(e)
    drain()
    el = time.perf_counter() - t0
    ts = [e0.elapsed_time(e) for e in evs]
    per = [round(ts[0], 2)] + [round(ts[i] - ts[i - 1], 2) for i in range(1, len(ts))]
    out.append((round(200 / el, 1), per))
for v, per in out:
    print(v, per)
