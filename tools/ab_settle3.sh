run() { name=$1; shift; env "$@" 2>/dev/null | tail -1 > gpurun_out/ab_$name.json
[ -s gpurun_out/ab_$name.json ] || { echo "$name: no output, stopping (sick box?)"; exit 9; }
python -c "import json,sys; d=json.load(open('gpurun_out/ab_$name.json')); print('$name', d['value'], d['ms_per_step'], d['value_spread']['values'])"; }
B="timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity"
for i in 1 2 3 4 5; do
run drain8_$i $B
run nodrain_$i LSFA_BENCH_SETTLE_DRAIN=0 $B
done
