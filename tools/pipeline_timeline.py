#!/usr/bin/env python
"""Where the pipelined frame loop's wall time goes, per hardware queue, from a rocprofv3 --kernel-trace directory (run on the GPU box):
for the last `frac` of the run - per queue: launches, busy time (union of its kernels), idle time inside its span and the largest idle
gaps with the kernels either side; over all queues: the time during which no kernel at all was running.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 30 --no-cpu-baseline --no-parity
    python3 tools/pipeline_timeline.py DIR [frac] [--chart MS [BUCKET_US]]

--chart: a text chart of the window's first MS milliseconds, one row per hardware queue, one character per BUCKET_US (default 250) microseconds:
the tenths of the bucket during which the queue had a kernel running ('.' = none, '#' = all of it).
"""
import collections
import csv
import glob
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('lsfa::', '').replace('convsplit::', '').replace('void ', '')
    return name.split('(')[0][:44]


def chart(byq, t0, ms, bucket_us):
    n = int(ms * 1e3 / bucket_us)
    print('\nchart: %d buckets of %d us from the window\'s start; per queue, tenths of the bucket with a kernel running' % (n, bucket_us))
    for q, ks in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        busy = [0.0] * n
        for s, e, _ in ks:
            a, b = (s - t0) / 1e3, (e - t0) / 1e3
            i = int(a // bucket_us)
            while i < n and i * bucket_us < b:
                lo, hi = max(a, i * bucket_us), min(b, (i + 1) * bucket_us)
                if hi > lo:
                    busy[i] += hi - lo
                i += 1
        line = ''.join('.' if x <= 0 else '#' if x >= 0.95 * bucket_us else str(min(9, int(10 * x / bucket_us))) for x in busy)
        print('queue %-3s %s' % (q, line))
    print('ms        ' + ''.join(('%-*d' % (int(1e3 / bucket_us), i)) for i in range(int(ms))))


def main(d, frac=0.5, chart_ms=0.0, bucket_us=250):
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    caps = [i for i, r in enumerate(rows) if 'det_cap_kernel' in r['Kernel_Name']]
    start = caps[int(len(caps) * (1 - frac))]
    images = lambda r: max(1, int(r.get('Grid_Size_X', 1)) // max(1, int(r.get('Workgroup_Size_X', 1))))      # one workgroup per image
    nframes = sum(images(rows[i]) for i in caps if i > start)
    rows = rows[start + 1:]
    t0, t1 = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
    wall = (t1 - t0) / 1e3
    print('window: %d frames, wall %.1f us (%.1f us per frame)' % (nframes, wall, wall / nframes))
    qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else 'Stream_Id'
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[qkey]].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    # all queues: union of busy intervals
    ev = sorted((s, e) for q in byq.values() for s, e, _ in q)
    cur_s, cur_e, union = ev[0][0], ev[0][1], 0
    for s, e in ev[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print('some kernel running: %.1f %% of the wall time; sum of kernel durations / wall = %.2f' %
          (100.0 * union / 1e3 / wall, sum(e - s for s, e in ev) / 1e3 / wall))
    for q, ks in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
        ks.sort()
        busy = sum(e - s for s, e, _ in ks) / 1e3
        span = (ks[-1][1] - ks[0][0]) / 1e3
        gaps = [(ks[i + 1][0] - ks[i][1], ks[i][2], ks[i + 1][2]) for i in range(len(ks) - 1)]
        idle = sum(max(g[0], 0) for g in gaps) / 1e3
        names = collections.Counter(short(k) for _, _, k in ks)
        print('\nqueue %s: %d launches, busy %.1f us (%.1f %% of wall), idle inside its span %.1f us; most frequent: %s' %
              (q, len(ks), busy, 100.0 * busy / wall, idle, ', '.join('%s x%d' % kv for kv in names.most_common(3))))
        hist = collections.Counter()
        for g, _, _ in gaps:
            us = g / 1e3
            hist['<2' if us < 2 else '2-5' if us < 5 else '5-20' if us < 20 else '20-100' if us < 100 else '>100'] += 1
        tot = {k: sum(max(g, 0) for g, _, _ in gaps if (lambda us: ('<2' if us < 2 else '2-5' if us < 5 else '5-20' if us < 20 else '20-100' if us < 100 else '>100'))(g / 1e3) == k) / 1e3
               for k in hist}
        print('   gaps between consecutive kernels (us): ' + ', '.join('%s: %d (%.0f us)' % (k, hist[k], tot[k]) for k in ('<2', '2-5', '5-20', '20-100', '>100') if k in hist))
        ctx = collections.defaultdict(lambda: [0, 0.0])
        for g, a, b in gaps:
            if g / 1e3 >= 20:
                c = ctx[(short(a), short(b))]
                c[0] += 1
                c[1] += g / 1e3
        for (a, b), (n, tt) in sorted(ctx.items(), key=lambda kv: -kv[1][1])[:6]:
            print('   %4d gaps >= 20 us, %8.0f us in all: after %-40s before %s' % (n, tt, a, b))


    if chart_ms > 0:
        chart(byq, t0, chart_ms, bucket_us)


if __name__ == '__main__':
    av = sys.argv[1:]
    cm, bu = 0.0, 250
    if '--chart' in av:
        i = av.index('--chart')
        extra = av[i + 1:]
        cm = float(extra[0])
        bu = int(extra[1]) if len(extra) > 1 else 250
        av = av[:i]
    main(av[0], float(av[1]) if len(av) > 1 else 0.5, cm, bu)
