for i in 1 2; do for v in "backbone,small,fuse,feat,stem,flow,nq" "backbone,small,fuse,feat,stem,flow,nq,h3"; do
echo "LSFA_OWN_CONV=$v"
LSFA_OWN_CONV=$v timeout 200 python tools/key_sections.py 2>/dev/null | grep -E "backbone|whole key|small net|whole non-key"
done; done
run() { name=$1; shift; env "$@" 2>/dev/null | tail -1 > gpurun_out/ab_$name.json
[ -s gpurun_out/ab_$name.json ] || { echo "$name: no output, stopping (sick box?)"; exit 9; }
python -c "import json,sys; d=json.load(open('gpurun_out/ab_$name.json')); print('$name', d['value'], d['value_spread']['values'], d['parity']['handwritten_stage_mismatches_on_gpu_inputs'], d['parity']['max_abs_dscore'], d['parity']['max_abs_dbox'], d['parity']['feature_rel_err_key_frame'])"; }
B="timeout 300 python bench.py --no-cpu-baseline"
for i in 1 2; do
run six_$i LSFA_OWN_CONV=backbone,small,fuse,feat,stem,flow,nq $B
run h3_$i LSFA_OWN_CONV=backbone,small,fuse,feat,stem,flow,nq,h3 $B
done
