mkdir -p gpurun_out/r18; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_hip_ops.py -m gpu -q --timeout 200 -p no:cacheprovider -x -k "conv" > gpurun_out/r18/hipops.log 2>&1
timeout 200 python tools/key_sections.py > gpurun_out/r18/key_sections.txt 2>&1
timeout 700 python -m pytest tests/test_graph_gpu.py tests/test_parity_fullres_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/r18/graph.log 2>&1
timeout 400 python bench.py --steps 100 > gpurun_out/r18/bench.json 2> gpurun_out/r18/bench.err
timeout 600 python tools/lab/conv_split_lab.py > gpurun_out/r18/conv_split_lab.txt 2>&1
tail -5 gpurun_out/r18/hipops.log; tail -5 gpurun_out/r18/graph.log; grep "backbone\|whole\|small" gpurun_out/r18/key_sections.txt; cut -c1-250 gpurun_out/r18/bench.json; cut -c1-40,150-290 gpurun_out/r18/conv_split_lab.txt
