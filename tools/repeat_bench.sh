for i in 1 2 3 4 5 6; do
timeout 150 python bench.py --no-cpu-baseline --no-parity --steps 40 > gpurun_out/rep_$i.out 2> gpurun_out/rep_$i.err; rc=$?
echo "run $i rc=$rc $(python -c "import json; d=json.loads(open('gpurun_out/rep_$i.out').read().strip().splitlines()[-1]); print(d['value'], d['value_spread']['values'])" 2>/dev/null)"
[ $rc -eq 0 ] || { tail -5 gpurun_out/rep_$i.err; break; }
done
