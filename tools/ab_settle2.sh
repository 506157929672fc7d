run() { name=$1; shift; "$@" 2>/dev/null | tail -1 > gpurun_out/ab_$name.json
[ -s gpurun_out/ab_$name.json ] || { echo "$name: no output, stopping (sick box?)"; exit 9; }
python -c "import json,sys; d=json.load(open('gpurun_out/ab_$name.json')); print('$name', d['value'], d['ms_per_step'], d['value_spread']['values'])"; }
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity"
for i in 1 2 3 4; do
run s25_$i $B
run s50_$i $B --settle-s 5
done
