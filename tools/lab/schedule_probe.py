#!/usr/bin/env python
"""When the pipelined loop's parts actually run, WITHOUT a profiler attached (a kernel trace slows the loop by 15 % and may reorder it):
bench.py's own run with timing events recorded around each pass of key fronts, each key frame's aggregation + tail and each segment pass, on the
streams they are queued on.  Prints the last groups' parts in start order, times relative to the first one shown.

    python3 tools/lab/schedule_probe.py [bench.py's flags]         (run on the GPU box)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench                                            # noqa: E402
from lsfa_amd.core import graphs                        # noqa: E402

LOG = []                                                # (label, start event, end event)


def timed(label, fn):
    def wrapped(self, *a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(self, *a, **k)
        e1.record()
        LOG.append((label, e0, e1))
        del LOG[:-400]
        return out
    return wrapped


graphs.KeyBank.run_front = timed('pass of fronts', graphs.KeyBank.run_front)
graphs.KeyBank.run_flow = timed('FlowNet pass', graphs.KeyBank.run_flow)
graphs.KeyLane.run_agg = timed('  aggregation', graphs.KeyLane.run_agg)
graphs.KeyLane.run_tail = timed('  key tail', graphs.KeyLane.run_tail)
graphs.FrameGraphs.cur_segment = timed('    segment pass', graphs.FrameGraphs.cur_segment)


def report():
    torch.cuda.synchronize()
    log = LOG[-120:]
    if not log:
        return
    base = log[0][1]
    rows = sorted((base.elapsed_time(e0), base.elapsed_time(e1), label) for label, e0, e1 in log)
    sys.stderr.write('\nschedule of the last %d parts (ms from the first; events on the parts\' own streams)\n' % len(rows))
    for a, b, label in rows:
        sys.stderr.write('%9.3f .. %9.3f  (%7.3f)  %s\n' % (a, b, b - a, label))


if __name__ == '__main__':
    sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:] + ['--no-cpu-baseline', '--no-parity', '--no-spread', '--no-frame-by-frame']
    real_profile = bench.Runner.eager_profile_batched

    def profile_after_report(self, *a, **k):            # the first thing bench.py does after its timed region
        report()
        return real_profile(self, *a, **k)
    bench.Runner.eager_profile_batched = profile_after_report
    bench.main()
