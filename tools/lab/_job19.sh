cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_graph_gpu.py -x -q -k "batched_segments" 2>&1 | tail -25 | cut -c1-220
for cfgs in "--segment 0 --key-group 1" "--segment 9 --key-group 1" "--segment 0 --key-group 3" "--segment 9 --key-group 2" "--segment 9 --key-group 3" "--segment 9 --key-group 3 --lookahead" "--segment 9 --key-group 4"; do
  timeout 400 python bench.py --steps 60 --no-cpu-baseline --no-parity $cfgs 2> gpurun_out/bench_err.txt | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print('$cfgs', d['value'], d['value_spread']['values'], d['roofline'].get('frac'), d['roofline'].get('achieved'))
except Exception as e:
    print('$cfgs', 'FAILED', e); print(open('gpurun_out/bench_err.txt').read()[-1500:])
"
done
