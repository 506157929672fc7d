"""per-interval key-stream cadence over ~9 s of uninterrupted bench.py load: when do the slow bursts happen?"""
import os, sys, time
sys.argv = ['bench.py', '--gpus', '1', '--steps', '20', '--warmup', '5', '--no-cpu-baseline', '--no-parity']
sys.path.insert(0, '.')
import torch
import bench
args = bench.parse()
r = bench.Runner(args, 0, 'cuda:0')
r.prime()
r.fg.flush(); torch.cuda.synchronize()
N = int(os.environ.get('N', '1500'))
evs = []
t0 = time.perf_counter()
e0 = torch.cuda.Event(enable_timing=True); e0.record(r.fg.s_key)
for s in range(N):
    r.step(s)
    e = torch.cuda.Event(enable_timing=True); e.record(r.fg.s_key); evs.append(e)
    if s % 64 == 63:
        evs[-32].synchronize()          # keep the host at most ~32 intervals ahead
r.fg.flush(); torch.cuda.synchronize()
ts = [e0.elapsed_time(e) for e in evs]
per = [ts[0]] + [ts[i] - ts[i - 1] for i in range(1, N)]
slow = [(i, round(ts[i] / 1e3, 3), round(per[i], 2)) for i in range(N) if per[i] > 6.5]
import statistics
print('intervals', N, 'median %.2f ms' % statistics.median(per), 'total %.2f s' % (ts[-1] / 1e3))
print('slow intervals (index, at second, ms):', slow)
