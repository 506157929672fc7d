cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_graph_gpu.py -x -q -k "nchw_to_nhwc or batched or library or key_frame or cur_frame or heads" 2>&1 | tail -5 | cut -c1-200
for cfgs in "" "--lanes 3" "--lanes 3 --lookahead" "--lookahead"; do
timeout 600 python bench.py --steps 60 --no-cpu-baseline --no-parity $cfgs 2> gpurun_out/bench_err.txt | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print('[$cfgs]', d['value'], d['value_spread']['values'], d['roofline'].get('frac'), d['roofline'].get('achieved'))
except Exception as e:
    print('[$cfgs]', 'FAILED', e); print(open('gpurun_out/bench_err.txt').read()[-1500:])
"
done
