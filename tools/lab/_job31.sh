cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | cut -c1-200
bash tools/profile_round.sh r4 > gpurun_out/profile_round_r4.log 2>&1
python - <<'PY'
import json
for f in ['bench_default_run.json','bench_frame_by_frame_run.json','bench_bf16_clips4_run.json','bench_bf16_clips1_run.json','bench_interval1_maps32_run.json']:
    try:
        d=json.loads(open('gpurun_out/r4/'+f).read().strip().splitlines()[-1])
        print(f, d['value'], d.get('value_spread',{}).get('values'), d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('achieved'), d['roofline'].get('traffic'), (d.get('cpu_baseline') or {}).get('value'), (d.get('parity') or {}).get('criterion_failures'))
    except Exception as e:
        print(f, 'ERR', e)
PY
