cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_hip_ops.py -x -q -k "det_postprocess or copy_many or stem or maxpool" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_graph_gpu.py -x -q -k "batched_segments or pinned_to_oracle" 2>&1 | tail -25
