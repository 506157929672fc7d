cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_ops.py tests/test_graph_gpu.py -x -q -k "rpn_head or heads or library or key_graph or batched" 2>&1 | tail -4 | cut -c1-200
timeout 200 python tools/key_sections.py 2>&1 | tail -7
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/cf9 -o t -- python3 tools/curframe_only.py 12 9 > /dev/null 2>&1
timeout 60 python3 tools/kernel_sequence.py /tmp/cf9 12 2>&1 | tail -16 | cut -c1-110
timeout 600 python bench.py --steps 60 --no-cpu-baseline --no-parity 2> gpurun_out/bench_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', d['value'], d['value_spread']['values'], d['roofline'].get('frac'), d['roofline'].get('achieved'))"
