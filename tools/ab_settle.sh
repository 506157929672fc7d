# the driver's own command line (--steps 20 --warmup 5), with and without the setup settle phase, on ONE fresh box
run() { name=$1; shift; "$@" 2>/dev/null | tail -1 > gpurun_out/ab_$name.json
[ -s gpurun_out/ab_$name.json ] || { echo "$name: no output, stopping (sick box?)"; exit 9; }
python -c "import json,sys; d=json.load(open('gpurun_out/ab_$name.json')); print('$name', d['value'], d['ms_per_step'], d['value_spread']['values'])"; }
B="timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-parity"
run settle0_a $B --settle-s 0
run settle1_a $B
run settle0_b $B --settle-s 0
run settle1_b $B
run settle3_b $B --settle-s 3
