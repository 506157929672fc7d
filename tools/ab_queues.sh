# A/B of whole-bench settings on ONE box: hardware queues x non-key lanes
run() { name=$1; shift; env "$@" 2>/dev/null | tail -1 > gpurun_out/ab_$name.json
[ -s gpurun_out/ab_$name.json ] || { echo "$name: no output, stopping (sick box?)"; exit 9; }
python -c "import json,sys; d=json.load(open('gpurun_out/ab_$name.json')); print('$name', d['value'], d['value_spread']['values'])"; }
B="timeout 300 python bench.py --no-cpu-baseline --no-parity"
for i in 1 2; do
run q4l2_$i $B
run q8l2_$i GPU_MAX_HW_QUEUES=8 $B
run q8l3_$i GPU_MAX_HW_QUEUES=8 $B --lanes 3
run q8l4_$i GPU_MAX_HW_QUEUES=8 $B --lanes 4
run q4l3_$i $B --lanes 3
run q4l1_$i $B --lanes 1
done
