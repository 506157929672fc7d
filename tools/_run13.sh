mkdir -p gpurun_out/r13; export TMPDIR=/tmp
timeout 300 python tools/lab/conv_split_lab.py --stamps > gpurun_out/r13/conv_split_stamps.txt 2>&1
timeout 300 python tools/lab/conv_split_lab.py > gpurun_out/r13/conv_split_lab.txt 2>&1
cat gpurun_out/r13/conv_split_stamps.txt; cut -c1-36,198-290 gpurun_out/r13/conv_split_lab.txt
