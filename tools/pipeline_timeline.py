#!/usr/bin/env python
"""Where the pipelined frame loop's wall time goes, per hardware queue, from a rocprofv3 --kernel-trace directory (run on the GPU box):
for the last `frac` of the run - per queue: launches, busy time (union of its kernels), idle time inside its span and the largest idle
gaps with the kernels either side; over all queues: the time during which no kernel at all was running.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 30 --no-cpu-baseline --no-parity
    python3 tools/pipeline_timeline.py DIR [frac]
"""
import collections
import csv
import glob
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('lsfa::', '').replace('convsplit::', '').replace('void ', '')
    return name.split('(')[0][:44]


def main(d, frac=0.5):
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


if __name__ == '__main__':
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)
