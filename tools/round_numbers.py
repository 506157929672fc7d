#!/usr/bin/env python
"""Print the numbers of a profile round (profiles/<round>/ or gpurun_out/<round>/) that DESIGN.md section 6 and the round's README quote, and
(--write-launch-counts) rebuild profiles/launch_counts.json from the round's kernel-sequence files.
    python tools/round_numbers.py profiles/r6 [--write-launch-counts]"""
import json
import os
import re
import sys

D = sys.argv[1] if len(sys.argv) > 1 else 'profiles/r6'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def line_of(name):
    try:
        txt = open(os.path.join(D, name)).read()
        return json.loads([l for l in txt.splitlines() if l.startswith('{')][-1])
    except (OSError, IndexError, ValueError):
        return None


def seq_total(name):
    """-> (sum of kernel durations in us, kernels) from a tools/kernel_sequence.py file"""
    try:
        m = re.search(r'sum of kernel durations: ([0-9.]+) us over (\d+) kernels', open(os.path.join(D, name)).read())
        return float(m.group(1)), int(m.group(2))
    except (OSError, AttributeError):
        return None, None


for f in ('bench_default_run', 'bench_driver_settings_run', 'bench_host_inputs_run', 'bench_frame_by_frame_run', 'bench_interval1_maps32_run',
          'bench_bf16_clips4_run', 'bench_bf16_clips1_run'):
    d = line_of(f + '.json')
    if d is None:
        print(f, 'missing')
        continue
    r = d.get('roofline', {})
    print('%-28s value %8.1f  spread %s  uploaded %s  frame-by-frame %s  exact %s' % (
        f, d['value'], (d.get('value_spread') or {}).get('values'), d.get('value_uploaded_inputs'), d.get('value_frame_by_frame'), d.get('value_exact_fp32')))
    print('    roofline: %s %s frac %s achieved %s traffic %s alg bytes %s launches %s avg_us %s' % (
        r.get('bound'), r.get('unit'), r.get('frac'), r.get('achieved'), r.get('traffic'), r.get('algorithmic_bytes_per_launch'), r.get('launches'), r.get('avg_us')))
    if 'cpu_baseline' in d:
        print('    cpu_baseline:', {k: d['cpu_baseline'].get(k) for k in ('value', 'unit', 'cores', 'kind')})
    p = d.get('parity')
    if p:
        print('    parity: failures %s  mismatches %s  map %s' % (p.get('criterion_failures'), p.get('handwritten_stage_mismatches_on_gpu_inputs'), p.get('map_vs_oracle')))
        wr = p.get('worst_frame_ratio') or p.get('worst_frame_ratios')
        if wr:
            print('    worst ratios:', wr)
for f in ('backbone6_kernel_sequence.txt', 'segment9_kernel_sequence.txt', 'flownet_kernel_sequence.txt', 'curframe_kernel_sequence.txt',
          'backbone_kernels_by_name.txt', 'backbone6_kernels_by_name.txt', 'backbone12_kernels_by_name.txt'):
    print('%-34s %s us over %s kernels' % ((f,) + seq_total(f)))
for f in ('conv_family_roofline.csv', 'key_batch_probe.txt', 'cur_batch_probe.txt', 'key_sections.txt', 'gpu_suite.txt'):
    try:
        print('--', f)
        print(''.join(open(os.path.join(D, f)).readlines()[-14:]).rstrip())
    except OSError:
        print('   missing')

if '--write-launch-counts' in sys.argv:
    lj = {}
    try:
        lj = json.load(open(os.path.join(D, 'bench_pipelined_timed_region_kernels_launches.json')))
    except (OSError, ValueError):
        pass
    rnd = os.path.basename(os.path.normpath(D))
    seg, nseg = seq_total('segment9_kernel_sequence.txt')
    cur, ncur = seq_total('curframe_kernel_sequence.txt')
    bb6, nbb6 = seq_total('backbone6_kernels_by_name.txt')
    bb12, nbb12 = seq_total('backbone12_kernels_by_name.txt')
    bb, nbb = seq_total('backbone_kernels_by_name.txt')
    fl, nfl = seq_total('flownet_kernel_sequence.txt')
    head = {"per_frame_mean_pipelined": lj.get("per_frame_mean"), "per_10_frame_interval_pipelined": lj.get("per_10_frame_interval"),
            "own_kernels_per_frame": lj.get("own_kernels_per_frame"), "library_gemm_per_frame": lj.get("library_gemm_or_conv_per_frame"),
            "gpu_busy_ms_per_frame": lj.get("gpu_busy_ms_per_frame"), "wall_ms_per_frame": lj.get("wall_ms_per_frame")}
    out = {"1000x600,interval=10,clips=1,f32": dict(
        head,
        segment_of_9_non_key_frames_eager={"launches": nseg, "kernel_time_us": seg, "note": "tools/curframe_only.py 12 9: ONE pass for the nine non-key frames of a segment"},
        non_key_frame_alone_eager={"launches": ncur, "kernel_time_us": cur, "note": "tools/curframe_only.py 30 (one frame per pass: the frame-by-frame pipeline)"},
        key_fronts_of_6_backbone_eager={"launches": nbb6, "kernel_time_us": bb6, "note": "tools/backbone_only.py 8 backbone 6: the image-only half of six key frames in one pass"},
        key_fronts_of_12_backbone_eager={"launches": nbb12, "kernel_time_us": bb12, "note": "tools/backbone_only.py 6 backbone 12: bench.py's default key group since r6"},
        key_frame_backbone_alone_eager={"launches": nbb, "kernel_time_us": bb},
        key_frame_flownet_alone_eager={"launches": nfl, "kernel_time_us": fl},
        source="profiles/%s/bench_pipelined_timed_region_kernels_launches.json (rocprofv3 --kernel-trace of bench.py --steps 30, batched pipeline), "
               "profiles/%s/segment9_kernel_sequence.txt, backbone6_kernels_by_name.txt, curframe_kernel_sequence.txt, backbone_kernels_by_name.txt, "
               "flownet_kernel_sequence.txt (tools/round_numbers.py --write-launch-counts)" % (rnd, rnd))}
    with open(os.path.join(ROOT, 'profiles', 'launch_counts.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('wrote profiles/launch_counts.json')
