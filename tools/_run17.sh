mkdir -p gpurun_out/r17; export TMPDIR=/tmp
timeout 600 python tools/lab/conv_split_lab.py > gpurun_out/r17/conv_split_lab.txt 2>&1
cut -c1-40,150-290 gpurun_out/r17/conv_split_lab.txt
