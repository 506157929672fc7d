#!/usr/bin/env python
"""Per-instantiation roofline table of the convolution family.

    python tools/conv_family_roofline.py <bench line json> <kernel stats csv> <pmc mfma-busy csv> <out csv>

  bench line json  bench.py's JSON line: roofline.by_kernel = {kernel instantiation: {calls, gflop, mbytes}} of the eagerly re-issued
                   intervals (which instantiation each lsfa_conv_fwd call runs on comes from lsfa_conv_plan_query, the FLOPs from the same
                   call's shape)
  kernel stats csv tools/summarize_prof.py trace ... of a rocprofv3 --kernel-trace run of the SAME loop (eager, `bench.py --no-graph`):
                   kernel,calls,total_us,avg_us,pct_of_busy
  pmc csv          tools/summarize_prof.py pmctable ...: kernel,dispatches,GRBM_GUI_ACTIVE,SQ_BUSY_CU_CYCLES,SQ_VALU_MFMA_BUSY_CYCLES

out: kernel, calls per profiled window, avg_us (kernel trace), GFLOP per call, TFLOP/s (= GFLOP per call / avg_us), fraction of the
matrix peak for its products per fp32 product, MB per call (algorithmic), MFMA duty (SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)).
A K-sliced launch ("+split_reduce") is listed with its ring kernel's time only; the reduce pass has its own row in the trace."""
import csv
import json
import re
import sys

PEAK = 2500.0


def short(name):
    m = re.search(r'(conv_ring_kernel<[^>]*>|conv_split_direct_kernel<[^>]*>|conv_split3x3_kernel<[^>]*>)', name)
    return m.group(1).replace(' ', '') if m else None


def main():
    line = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
    byk = (line.get('roofline') or {}).get('by_kernel') or (line.get('roofline_mfma_kernel') or {}).get('by_kernel') or {}
    stats, pmc = {}, {}
    for r in csv.DictReader(l for l in open(sys.argv[2]) if not l.startswith('#')):
        k = short(r['kernel'])
        if k:
            stats[k] = r
    for r in csv.DictReader(open(sys.argv[3])):
        k = short(r['kernel'])
        if k:
            pmc[k] = r
    rows = []
    # a K-sliced launch runs the same ring instantiation (+ a reduce pass, which has its own row in the trace): one row per instantiation
    merged = {}
    for name, v in byk.items():
        base = name.replace(' +split_reduce', '')
        m = merged.setdefault(base, {'calls': 0, 'gflop': 0.0, 'mbytes': 0.0, 'sliced_calls': 0})
        m['calls'] += v['calls']; m['gflop'] += v['gflop']; m['mbytes'] += v['mbytes']
        if name != base:
            m['sliced_calls'] += v['calls']
    for name, v in merged.items():
        k = short(name)
        if not k:
            continue
        # rocprofv3 prints defaulted template arguments too: <NT, PC, ST, SP, AF, WV>
        st = stats.get(k)
        pc = int(k.split('<')[1].split(',')[1].rstrip('>')) if 'ring' in k else int(k.split('<')[1].rstrip('>').split(',')[-1])
        products = {1: 1, 2: 3, 3: 6}[pc]
        gf = v['gflop'] / max(v['calls'], 1)
        avg = float(st['avg_us']) if st else None
        tf = gf / avg * 1e3 if avg else None
        pm = pmc.get(k)
        duty = (float(pm['SQ_VALU_MFMA_BUSY_CYCLES']) / (4.0 * float(pm['SQ_BUSY_CU_CYCLES']))) if pm and float(pm['SQ_BUSY_CU_CYCLES']) > 0 else None
        rows.append((name, v['calls'], avg, gf, tf, (tf / (PEAK / products)) if tf else None, v['mbytes'] / max(v['calls'], 1), duty, v['sliced_calls']))
    rows.sort(key=lambda r: -(r[1] * (r[2] or 0)))
    with open(sys.argv[4], 'w') as o:
        o.write('# %s\n' % (line.get('config', {}).get('workload', '')))
        o.write('kernel,calls,avg_us,gflop_per_call,tflops,frac_of_peak_for_its_products,algorithmic_mb_per_call,mfma_duty,of_which_k_sliced_calls\n')
        for r in rows:
            o.write('"%s",%d,%s,%.3f,%s,%s,%.2f,%s,%d\n' % (r[0], r[1], '%.2f' % r[2] if r[2] else '', r[3], '%.1f' % r[4] if r[4] else '',
                                                           '%.3f' % r[5] if r[5] else '', r[6], '%.3f' % r[7] if r[7] is not None else '', r[8]))
        # the trace window holds several repetitions of the loop; the rows above count ONE (by_kernel): scale the reduce passes' counts alike
        reps = sorted(float(stats[short(n)]['calls']) / v['calls'] for n, v in merged.items() if short(n) in stats and v['calls'])
        rep = reps[len(reps) // 2] if reps else 1.0
        red = [r for r in csv.DictReader(l for l in open(sys.argv[2]) if not l.startswith('#')) if 'split_reduce' in r['kernel']]
        for r in red:
            o.write('"%s",%.1f,%s,,,,,,\n' % (r['kernel'].split('(')[0].replace('lsfa::convsplit::', ''), float(r['calls']) / rep, r['avg_us']))
        tot_gf = sum(v['gflop'] for v in byk.values())
        tot_us = sum(r[1] * r[2] for r in rows if r[2]) + sum(float(r['calls']) / rep * float(r['avg_us']) for r in red)
        o.write('# all listed (reduce passes included): %.1f GFLOP in %.1f us of kernel time = %.1f TFLOP/s\n' % (tot_gf, tot_us, tot_gf / tot_us * 1e3 if tot_us else 0.0))


if __name__ == '__main__':
    main()
