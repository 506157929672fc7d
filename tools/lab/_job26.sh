cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_fullres_gpu.py tests/test_graph_gpu.py -x -q -k "1000x600 or evicts or library" 2>&1 | tail -8 | cut -c1-220
timeout 900 bash tools/lab/drop_product_check.sh > gpurun_out/drop_product_check.txt 2>&1; tail -12 gpurun_out/drop_product_check.txt | cut -c1-200
bash tools/profile_round.sh r4 > gpurun_out/profile_round_r4.log 2>&1
ls gpurun_out/r4 | wc -l
python - <<'PY'
import json
for f in ['bench_default_run.json','bench_frame_by_frame_run.json','bench_bf16_clips4_run.json','bench_bf16_clips1_run.json','bench_interval1_maps32_run.json']:
    try:
        d=json.loads(open('gpurun_out/r4/'+f).read().strip().splitlines()[-1])
        print(f, d['value'], d.get('value_spread',{}).get('values'), d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('achieved'), (d.get('parity') or {}).get('criterion_failures'))
    except Exception as e:
        print(f, 'ERR', e)
PY
cat gpurun_out/r4/pipeline_timeline.txt | head -40 | cut -c1-200
