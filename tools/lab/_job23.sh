cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | cut -c1-220
for cfgs in "" "--dtype bf16 --clips 4" "--dtype bf16 --clips 4 --segment 0 --key-group 1" "--dtype bf16"; do
timeout 600 python bench.py --steps 40 --no-cpu-baseline --no-parity $cfgs 2> gpurun_out/bench_err.txt | python -c "
import json,sys
t=sys.stdin.read().strip().splitlines()
try:
    d=json.loads(t[-1]); print('[$cfgs]', d['value'], d['value_spread']['values'], d['roofline'].get('frac'), d['roofline'].get('achieved'))
except Exception as e:
    print('[$cfgs]', 'FAILED', e); print(open('gpurun_out/bench_err.txt').read()[-1500:])
"
done
