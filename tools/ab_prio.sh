# A/B of whole-bench settings on ONE box (boxes differ by +-4 %): name=ENV pairs, two rounds
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-parity 2>/dev/null | tail -1 > gpurun_out/ab_$name.json
[ -s gpurun_out/ab_$name.json ] || { echo "$name: no output, stopping (sick box?)"; exit 9; }
python -c "import json,sys; d=json.load(open('gpurun_out/ab_$name.json')); print('$name', d['value'], d['value_spread']['values'])"; }
for i in 1 2; do
run probe$i LSFA_STREAM_LAYOUT=probe
run plain$i LSFA_STREAM_LAYOUT=plain
run keyhi$i LSFA_STREAM_LAYOUT=plain LSFA_STREAM_PRIO=key
run lanehi$i LSFA_STREAM_LAYOUT=plain LSFA_STREAM_PRIO=lanes
run keyflowhi$i LSFA_STREAM_LAYOUT=plain LSFA_STREAM_PRIO=key,flow
done
