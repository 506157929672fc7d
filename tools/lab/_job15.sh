cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
hipcc -O3 --offload-arch=gfx950 tools/lab/grid_barrier_lab.hip -o /tmp/gbl && timeout 120 /tmp/gbl > gpurun_out/grid_barrier_lab.txt 2>&1
tail -30 gpurun_out/grid_barrier_lab.txt
timeout 300 python tools/lab/key_batch_probe.py 2>&1 | tail -6 | tee gpurun_out/key_batch_probe.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 bench.py --steps 30 --no-cpu-baseline --no-parity > /dev/null 2>&1
python3 tools/pipeline_timeline.py /tmp/tl 0.5 > gpurun_out/pipeline_timeline.txt 2>&1
cat gpurun_out/pipeline_timeline.txt
