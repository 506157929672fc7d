for i in 1 2; do
for v in gather auto; do
LSFA_WARP_VARIANT=$v timeout 300 python bench.py --no-cpu-baseline --no-parity 2>/dev/null | tail -1 > gpurun_out/ab_$v$i.json
python -c "import json,sys; d=json.load(open('gpurun_out/ab_$v$i.json')); print('$v', d['value'], d['value_spread']['values'], d['roofline_hbm_kernel']['avg_us'])"
done; done
