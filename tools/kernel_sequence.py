#!/usr/bin/env python
"""For a rocprofv3 kernel trace of a script that repeats ONE launch sequence n times (tools/backbone_only.py): the kernels of one
steady-state repetition in launch order with their durations averaged over the last repetitions.
usage: kernel_sequence.py <trace_dir> <repetitions> [--by-name]     (--by-name: one line per kernel name, time summed over the
repetition, for sequences of hundreds of launches such as the backbone)"""
import csv
import glob
import os
import sys


def main():
    d, n = sys.argv[1], int(sys.argv[2])
    f = [p for p in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)][0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'] for r in rows]
    # the period of the steady state: the shortest p with the last p names equal to the p before them (the first repetition
    # also runs one-off preparation kernels, so len(rows) / n is not it)
    # (three equal periods, at least five launches long: a run of same-named convolutions is not a period)
    per = next(p for p in range(5, len(rows) // 3) if names[-p:] == names[-2 * p:-p] == names[-3 * p:-2 * p])
    reps = max(1, min(n // 2, len(rows) // per - 1))
    tail = rows[len(rows) - per * reps:]
    tot = 0.0
    if '--by-name' in sys.argv:
        agg = {}
        for i in range(per):
            name = tail[i]['Kernel_Name']
            dur = sum((int(tail[k * per + i]['End_Timestamp']) - int(tail[k * per + i]['Start_Timestamp'])) for k in range(reps)) / reps / 1e3
            a = agg.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += dur
            tot += dur
        for name, (cnt, dur) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            short = name.replace('(anonymous namespace)::', '').replace('lsfa::convsplit::', '').replace('void ', '').split('(')[0][:90]
            print('%-92s %4d launches %9.1f us  %5.1f %%' % (short, cnt, dur, 100 * dur / tot))
        print('sum of kernel durations: %.1f us over %d kernels (%d repetitions averaged)' % (tot, per, reps))
        return
    for i in range(per):
        name = tail[i]['Kernel_Name']
        dur = sum((int(tail[k * per + i]['End_Timestamp']) - int(tail[k * per + i]['Start_Timestamp'])) for k in range(reps)) / reps / 1e3
        same = all(tail[k * per + i]['Kernel_Name'] == name for k in range(reps))
        tot += dur
        short = name.replace('(anonymous namespace)::', '').replace('lsfa::convsplit::', '').replace('void ', '').split('(')[0][:60]
        print('%3d %-62s %8.1f us  grid %s%s' % (i, short, dur, tail[i].get('Grid_Size', '?'), '' if same else '  (sequence differs between repetitions)'))
    print('sum of kernel durations: %.1f us over %d kernels (%d repetitions averaged)' % (tot, per, reps))


if __name__ == '__main__':
    main()
