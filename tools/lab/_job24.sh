cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_graph_gpu.py tests/test_multirank_gpu.py tests/test_configs_gpu.py -x -q 2>&1 | tail -15 | cut -c1-220
