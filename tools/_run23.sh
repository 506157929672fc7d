mkdir -p gpurun_out/r23; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/r23/gpu_suite.log 2>&1
tail -5 gpurun_out/r23/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r23/smoke.log 2>&1; tail -3 gpurun_out/r23/smoke.log
