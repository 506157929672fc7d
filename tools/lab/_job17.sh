cd $GRAFT_REPO_ROOT
timeout 300 python tools/lab/cur_batch_probe.py 2>&1 | tail -6 | tee gpurun_out/cur_batch_probe.txt
timeout 200 python tools/key_sections.py 2>&1 | tail -7
