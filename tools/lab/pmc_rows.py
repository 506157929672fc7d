#!/usr/bin/env python
"""counter_collection.csv files of rocprofv3 --pmc passes over conv_trace_probe.py -> per (kernel, grid): mean of each counter"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(path)):
            n = r['Kernel_Name']
            if not ('conv_ring' in n or 'reduce' in n or 'direct' in n):
                continue
            short = n.split('(')[0].replace('lsfa::convsplit::', '').replace('void ', '')[:40]
            acc[(short, r.get('Grid_Size', '?'))][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, grid), cs in sorted(acc.items()):
    print("%s grid %s" % (k, grid))
    for c, v in sorted(cs.items()):
        v = v[1:] if len(v) > 2 else v        # the first launch of a plan is cold
        print("    %-34s %14.0f   (n=%d)" % (c, sum(v) / len(v), len(v)))
