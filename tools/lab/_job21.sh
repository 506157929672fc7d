cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/cf9 -o t -- python3 tools/curframe_only.py 20 9 > gpurun_out/cf9.log 2>&1
tail -5 gpurun_out/cf9.log
timeout 60 python3 tools/kernel_sequence.py /tmp/cf9 20 > gpurun_out/segment9_kernel_sequence.txt 2>&1; cat gpurun_out/segment9_kernel_sequence.txt | cut -c1-150
timeout 1500 python3 tools/lab/conv_ring_lab.py --quick --pieces 2 --batch 3 --shapes "res4,res5,res3,res2,feat" > gpurun_out/conv_ring_lab_batch3.txt 2>&1
grep -E "^##|plan's choice|best plans" gpurun_out/conv_ring_lab_batch3.txt | cut -c1-220
