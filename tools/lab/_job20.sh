cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/cf9 -o t -- python3 tools/curframe_only.py 20 9 > /dev/null 2>&1
timeout 60 python3 tools/kernel_sequence.py /tmp/cf9 20 > gpurun_out/segment9_kernel_sequence.txt 2>&1; cat gpurun_out/segment9_kernel_sequence.txt | cut -c1-150
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/bb3 -o t -- python3 tools/backbone_only.py 10 backbone 3 > /dev/null 2>&1
timeout 60 python3 tools/kernel_sequence.py /tmp/bb3 10 --by-name > gpurun_out/backbone3_by_name.txt 2>&1; cat gpurun_out/backbone3_by_name.txt | cut -c1-150
timeout 60 python3 tools/kernel_sequence.py /tmp/bb3 10 > gpurun_out/backbone3_sequence.txt 2>&1; tail -3 gpurun_out/backbone3_sequence.txt
