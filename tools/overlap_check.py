#!/usr/bin/env python
"""Do kernels of the forked small-net branch overlap the main branch inside graph replays?
Reads a rocprofv3 kernel trace; reports the fraction of kernel time that overlaps another kernel."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:50], r['Queue_Id'], r['Stream_Id']) for r in csv.DictReader(open(f))]
rows.sort()
caps = [i for i, r in enumerate(rows) if "det_cap_kernel" in r[2]]
rows = rows[caps[-41]:] if len(caps) > 45 else rows[len(rows) // 2:]   # the last 40 frames
busy = sum(e - s for s, e, _, _, _ in rows)
ov = 0
last_end = 0
for s, e, *_ in rows:
    if s < last_end:
        ov += min(e, last_end) - s
    last_end = max(last_end, e)
print('kernels', len(rows), 'busy ms %.2f' % (busy / 1e6), 'overlapped ms %.2f (%.1f%%)' % (ov / 1e6, 100.0 * ov / busy))
print('queues', sorted(set(r[3] for r in rows)), 'streams', sorted(set(r[4] for r in rows))[:10])
