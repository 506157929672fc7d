cd $GRAFT_REPO_ROOT
timeout 300 python tools/lab/cur_batch_probe.py 2>&1 | tail -6 | tee gpurun_out/cur_batch_probe.txt
timeout 200 python -m pytest tests/test_hip_ops.py -x -q -k "chain" 2>&1 | tail -5
LSFA_SMALL_NET_CHAIN=1 timeout 200 python tools/key_sections.py 2>&1 | tail -2
LSFA_SMALL_NET_CHAIN=0 timeout 200 python tools/key_sections.py 2>&1 | tail -2
