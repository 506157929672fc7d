cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -k "conv" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_graph_gpu.py -x -q -k "evicts or batched_segments" 2>&1 | tail -12 | cut -c1-200
timeout 300 python tools/lab/key_batch_probe.py 2>&1 | tail -4
timeout 400 python bench.py --steps 60 --no-cpu-baseline --no-parity 2> gpurun_out/bench_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['value_spread']['values'], d['roofline'].get('frac'), d['roofline'].get('achieved'))"
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/cf9 -o t -- python3 tools/curframe_only.py 12 9 > gpurun_out/cf9.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/cf9/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
print(len(rows), 'rows')
short = lambda n: n.replace('(anonymous namespace)::', '').replace('lsfa::convsplit::', '').replace('void ', '').split('(')[0][:60]
for r in rows[-140:]:
    print('%-62s %8.1f' % (short(r['Kernel_Name']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
