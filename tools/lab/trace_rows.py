#!/usr/bin/env python
"""kernel_trace.csv of conv_trace_probe.py -> one line per launch: name, duration (us), grid, LDS, in order"""
import csv
import glob
import sys

path = sorted(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for r in rows:
    n = r['Kernel_Name']
    if 'conv_ring' in n or 'reduce' in n or 'direct' in n or 'FillFunctor' in n:
        short = n.split('(')[0].replace('lsfa::convsplit::', '').replace('void ', '')[:60]
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        print("%-62s %8.2f us  grid %s  wg %s  lds %s" % (short, dur, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')), r.get('LDS_Block_Size', '?')))
