#!/usr/bin/env python
"""For a rocprofv3 kernel trace of the pipelined bench: the idle gaps of the key queue (the one running most conv_ring launches) longer than
1 ms in the second half of the run, and what the other queues ran during each (first / last kernels, busy time)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[len(rows) // 2:]
qk = 'Queue_Id'
byq = collections.defaultdict(list)
for r in rows:
    byq[r[qk]].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][-40:]))
key = max(byq, key=lambda q: sum(1 for k in byq[q] if 'conv_ring' in k[2]) * (1 if sum(e - s for s, e, _ in byq[q]) > 0 else 0))
# the key queue: most ring launches AND the longest kernels (the backbone passes)
key = max(byq, key=lambda q: sum(e - s for s, e, n in byq[q] if 'conv_ring' in n))
ks = byq[key]
shown = 0
for i in range(len(ks) - 1):
    g0, g1 = ks[i][1], ks[i + 1][0]
    if g1 - g0 < 1e6:
        continue
    print('key queue %s idle %.2f ms after %s before %s' % (key, (g1 - g0) / 1e6, ks[i][2], ks[i + 1][2]))
    for q, v in byq.items():
        if q == key:
            continue
        inside = [(s, e, n) for s, e, n in v if e > g0 and s < g1]
        if not inside:
            print('   queue %s: nothing' % q)
            continue
        busy = sum(min(e, g1) - max(s, g0) for s, e, n in inside) / 1e6
        print('   queue %s: %d kernels, busy %.2f ms; first %s (+%.2f ms), last %s (ends %+.2f ms vs the gap end)' %
              (q, len(inside), busy, inside[0][2], (inside[0][0] - g0) / 1e6, inside[-1][2], (inside[-1][1] - g1) / 1e6))
    shown += 1
    if shown >= 6:
        break

# how many queues run kernels at the same time (share of the second half of the run)
ev = []
for q, v in byq.items():
    # merge a queue's back-to-back kernels into busy intervals
    v.sort()
    cs, ce = v[0][0], v[0][1]
    for s, e, _ in v[1:]:
        if s - ce > 2000:          # > 2 us apart: the queue was idle
            ev += [(cs, 1, q), (ce, -1, q)]
            cs, ce = s, e
        else:
            ce = max(ce, e)
    ev += [(cs, 1, q), (ce, -1, q)]
ev.sort()
hist = collections.Counter()
active, last = 0, ev[0][0]
for t, d_, q in ev:
    hist[active] += t - last
    active += d_
    last = t
tot = float(sum(hist.values()))
print('queues running kernels at the same time: ' + ', '.join('%d: %.1f %%' % (k, 100.0 * hist[k] / tot) for k in sorted(hist)))
