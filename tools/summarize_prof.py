#!/usr/bin/env python
"""Condense rocprofv3 CSV output into small summaries (run on the GPU box; traces are too big to ship).

  kernel-trace dir -> per-kernel calls / total / avg over the LAST `frac` of the run (timed region)
  pmc dir          -> per-kernel mean counter value for kernels matching a substring
"""
import csv, glob, sys, collections, json


def trace_summary(d, out, tail_frac=0.5, top=60):
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)
    if not f:
        return
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    caps = [i for i, r in enumerate(rows) if 'det_cap_kernel' in r['Kernel_Name']]

    def images(r):      # det_cap_kernel runs one workgroup per image: a key frame's launch is one frame, a batched segment's is its frames
        try:
            return max(1, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])))
        except (KeyError, ValueError):
            return 1
    if len(caps) > 4:
        start = caps[int(len(caps) * (1 - tail_frac))]
        nframes = sum(images(rows[i]) for i in caps if i > start)
        rows = rows[start + 1:]
    else:
        nframes = 0
        rows = rows[int(len(rows) * (1 - tail_frac)):]
    t0, t1 = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        a = agg[r['Kernel_Name']]
        a[0] += 1
        a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    busy = sum(v[1] for v in agg.values())
    with open(out, 'w') as o:
        o.write('# window: last %.0f%% of the run = %d frames, wall %.3f ms, GPU busy %.3f ms, %d kernel launches\n' %
                (tail_frac * 100, nframes, (t1 - t0) / 1e6, busy / 1e6, len(rows)))
        o.write('kernel,calls,total_us,avg_us,pct_of_busy\n')
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
            o.write('"%s",%d,%.1f,%.2f,%.2f\n' % (k[:140], v[0], v[1] / 1e3, v[1] / v[0] / 1e3, 100.0 * v[1] / busy))
    if nframes:
        # launches per frame over the same window -> what bench.py reports as `kernel_launches` (profiles/launch_counts.json)
        own = sum(v[0] for k, v in agg.items() if 'lsfa' in k or '(anonymous namespace)' in k)
        lib = sum(v[0] for k, v in agg.items() if k.startswith('Cijk') or 'miopen' in k.lower() or 'ck::' in k or 'igemm' in k)
        with open(out.replace('.csv', '') + '_launches.json', 'w') as o:
            json.dump({"frames_in_window": nframes, "kernel_launches_in_window": len(rows),
                       "per_frame_mean": round(len(rows) / float(nframes), 1), "per_10_frame_interval": round(10.0 * len(rows) / nframes, 1),
                       "own_kernels_per_frame": round(own / float(nframes), 1), "library_gemm_or_conv_per_frame": round(lib / float(nframes), 1),
                       "gpu_busy_ms_per_frame": round(busy / 1e6 / nframes, 4), "wall_ms_per_frame": round((t1 - t0) / 1e6 / nframes, 4),
                       "own_kernel_share_of_busy": round(sum(v[1] for k, v in agg.items() if 'lsfa' in k or '(anonymous namespace)' in k) / float(busy), 4),
                       "source": "rocprofv3 --kernel-trace of `bench.py --steps 30` (pipelined hipGraph replay), last %.0f%% of the run" % (tail_frac * 100)}, o, indent=1)


def pmc_summary(d, match):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f:
        return None
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if match in r['Kernel_Name']:
            vals[r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: dict(n=len(v), mean=sum(v) / len(v), min=min(v), max=max(v)) for k, v in vals.items()}


def pmc_table(d, out, top=45, last_frames=20):
    """every kernel: dispatches and the SUM of each counter over its dispatches, sorted by GRBM_GUI_ACTIVE."""
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f:
        return
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    rows = list(csv.DictReader(open(f[0])))
    caps = sorted({int(r['Dispatch_Id']) for r in rows if 'det_cap_kernel' in r['Kernel_Name']})
    first = caps[-(last_frames + 1)] if len(caps) > last_frames else -1     # skip warm-up / MIOpen find dispatches
    for r in rows:
        if int(r['Dispatch_Id']) <= first:
            continue
        k = r['Kernel_Name'][:120]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        disp[k].add(r['Dispatch_Id'])
    names = sorted({c for v in agg.values() for c in v})
    key = 'GRBM_GUI_ACTIVE' if 'GRBM_GUI_ACTIVE' in names else names[0]
    with open(out, 'w') as o:
        o.write('kernel,dispatches,' + ','.join(names) + '\n')
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get(key, 0))[:top]:
            o.write('"%s",%d,%s\n' % (k, len(disp[k]), ','.join('%.0f' % v.get(c, 0) for c in names)))


def conv_traffic(fetch_dir, write_dir, out_json, key):
    """HBM bytes of the split-convolution kernel family per lsfa_conv_fwd call, from two PMC passes (FETCH_SIZE, WRITE_SIZE: they
    cannot share one) over the same eager loop: (2 x sum FETCH_SIZE KiB [the gfx950 correction] + sum WRITE_SIZE KiB) x 1024 over the
    family's dispatches / dispatches of its main kernels (a call = one main kernel + at most one reduce / fix-up pass).  Merged
    into `out_json` (profiles/traffic.json, what bench.py reads) under `key`."""
    main = ('conv_ring_kernel', 'conv_split_direct_kernel', 'conv_split3x3_kernel')

    def collect(d, cname):
        f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f[0])):
            if r['Counter_Name'] == cname and 'convsplit::' in r['Kernel_Name']:
                k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('lsfa::convsplit::', '')
                agg[k][0] += float(r['Counter_Value'])
                agg[k][1] += 1
        return agg
    fe, wr = collect(fetch_dir, 'FETCH_SIZE'), collect(write_dir, 'WRITE_SIZE')
    calls_f = sum(v[1] for k, v in fe.items() if k.startswith(main) and 'fixup' not in k)
    calls_w = sum(v[1] for k, v in wr.items() if k.startswith(main) and 'fixup' not in k)
    fetch_per_call = 2.0 * 1024 * sum(v[0] for v in fe.values()) / max(calls_f, 1)
    write_per_call = 1024.0 * sum(v[0] for v in wr.values()) / max(calls_w, 1)
    entry = {"hbm_bytes_per_launch": int(fetch_per_call + write_per_call), "fetch_bytes_per_launch": int(fetch_per_call),
             "write_bytes_per_launch": int(write_per_call), "calls_in_fetch_pass": calls_f, "calls_in_write_pass": calls_w,
             "per_kernel_avg_bytes": {k: {"dispatches": fe[k][1], "fetch": int(2048 * fe[k][0] / max(fe[k][1], 1)),
                                          "write": int(1024 * wr[k][0] / max(wr[k][1], 1)) if k in wr else None} for k in sorted(fe)},
             "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --no-graph` (the batched pipeline issued eagerly); FETCH_SIZE x2 "
                       "(gfx950), KiB units; a launch = one lsfa_conv_fwd call (main kernel + its reduce / fix-up pass)"}
    try:
        d = json.load(open(out_json))
    except (OSError, ValueError):
        d = {}
    d[key] = entry
    json.dump(d, open(out_json, 'w'), indent=1)
    print(json.dumps(entry, indent=1))


if __name__ == '__main__':
    mode = sys.argv[1]
    if mode == 'convtraffic':
        conv_traffic(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
    elif mode == 'pmctable':
        pmc_table(sys.argv[2], sys.argv[3])
    elif mode == 'trace':
        trace_summary(sys.argv[2], sys.argv[3], float(sys.argv[4]) if len(sys.argv) > 4 else 0.5)
    else:
        print(json.dumps(pmc_summary(sys.argv[2], sys.argv[3])))
