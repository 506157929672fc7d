cd $GRAFT_REPO_ROOT/tools/lab
timeout 600 python lda_pad_probe.py 3 2>&1 | tail -8 | tee ../../gpurun_out/lda_pad_probe.txt
timeout 600 python lda_pad_probe.py 1 2>&1 | tail -8 | tee -a ../../gpurun_out/lda_pad_probe.txt
