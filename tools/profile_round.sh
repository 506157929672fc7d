#!/bin/bash
# The rocprofv3 passes behind profiles/<round>/ (run on the GPU box: gpurun -- bash tools/profile_round.sh r2).
# Raw traces stay on the box; the condensed summaries land in gpurun_out/<round>/ and are copied to profiles/<round>/.
# PMC passes never share a run with --sys-trace / --runtime-trace (the pool refuses that combination).
set -u
R=${1:-r6}
OUT=gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
PY=python3

# 1. the bench lines (default = configs[1], batched passes; the same frame by frame; configs[4]; configs[2])
timeout 600 $PY bench.py --steps 100 > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err
timeout 300 $PY bench.py --steps 20 --warmup 5 > $OUT/bench_driver_settings_run.json 2> /dev/null                                   # the driver's command line
timeout 300 $PY bench.py --steps 100 --host-inputs --no-cpu-baseline --no-parity > $OUT/bench_host_inputs_run.json 2> /dev/null          # the PCIe-inclusive rate as `value`
timeout 400 $PY bench.py --steps 60 --segment 0 --key-group 1 --no-cpu-baseline > $OUT/bench_frame_by_frame_run.json 2> /dev/null
timeout 400 $PY bench.py --interval 1 --maps-per-launch 32 --steps 60 > $OUT/bench_interval1_maps32_run.json 2> $OUT/bench_interval1_maps32_run.err
timeout 600 $PY bench.py --dtype bf16 --clips 4 --steps 40 > $OUT/bench_bf16_clips4_run.json 2> $OUT/bench_bf16_clips4_run.err
timeout 400 $PY bench.py --dtype bf16 --steps 60 --no-cpu-baseline --no-parity > $OUT/bench_bf16_clips1_run.json 2> /dev/null

# 2. kernel trace of the pipelined timed region (default configuration)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pipe -o t -- $PY bench.py --steps 30 --no-cpu-baseline --no-parity --no-frame-by-frame > $OUT/trace_pipe.log 2>&1
$PY tools/summarize_prof.py trace $OUT/trace_pipe $OUT/bench_pipelined_timed_region_kernels.csv 0.6     # also writes ..._kernels_launches.json
cp $(find $OUT/trace_pipe -name "*kernel_stats.csv" | head -1) $OUT/bench_pipelined_kernel_stats_whole_run.csv 2>/dev/null
rm -rf $OUT/trace_pipe

# 3. matrix-pipe duty per kernel, serial eager loop (counters only: no trace domains next to --pmc)
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -o t -- $PY bench.py --steps 12 --warmup 3 --settle-s 0 --no-graph --no-cpu-baseline --no-parity --no-spread --no-frame-by-frame > $OUT/pmc_mfma.log 2>&1
$PY tools/summarize_prof.py pmctable $OUT/pmc_mfma $OUT/bench_eager_pmc_mfma_busy.csv
rm -rf $OUT/pmc_mfma

# 3b. r5: the per-instantiation roofline table of the convolution family: FLOPs per kernel instantiation from the
#     binding (lsfa_conv_plan_query), durations from a kernel trace and MFMA-busy from a PMC pass of the SAME serial eager loop
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_eager -o t -- $PY bench.py --eager-loop 6 --settle-s 0 > $OUT/bench_eager_loop.json 2> $OUT/bench_eager_loop.err
$PY tools/summarize_prof.py trace $OUT/trace_eager $OUT/bench_eager_loop_kernels.csv 0.8
rm -rf $OUT/trace_eager
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_eager -o t -- $PY bench.py --eager-loop 6 --settle-s 0 > $OUT/pmc_eager.log 2>&1
$PY tools/summarize_prof.py pmctable $OUT/pmc_eager $OUT/bench_eager_loop_pmc_mfma_busy.csv
rm -rf $OUT/pmc_eager
$PY tools/conv_family_roofline.py $OUT/bench_eager_loop.json $OUT/bench_eager_loop_kernels.csv $OUT/bench_eager_loop_pmc_mfma_busy.csv $OUT/conv_family_roofline.csv

# 4. HBM traffic of the streaming kernels (FETCH_SIZE and WRITE_SIZE cannot share a pass) -> traffic.json
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o t -- $PY tools/traffic_probe.py run > $OUT/traffic_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o t -- $PY tools/traffic_probe.py run > $OUT/traffic_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc_trace -o t -- $PY tools/traffic_probe.py run > $OUT/traffic_trace.log 2>&1
$PY tools/traffic_probe.py summarize $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_trace $OUT/traffic.json > $OUT/traffic_summary.log 2>&1
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_trace
# 4b. HBM traffic of the split-convolution family per call (the `roofline.traffic` of the default bench line), same two-pass rule
EAGER="bench.py --steps 12 --warmup 12 --settle-s 0 --no-graph --no-cpu-baseline --no-parity --no-spread --no-frame-by-frame"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_cfetch -o t -- $PY $EAGER > $OUT/conv_traffic_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_cwrite -o t -- $PY $EAGER > $OUT/conv_traffic_write.log 2>&1
$PY tools/summarize_prof.py convtraffic $OUT/pmc_cfetch $OUT/pmc_cwrite $OUT/traffic.json "conv_split:1000x600,interval=10,f32" > $OUT/conv_traffic_summary.log 2>&1
rm -rf $OUT/pmc_cfetch $OUT/pmc_cwrite
# 4c. r6: the same for configs[2] (bf16, four clips): VERDICT r5 found `roofline.traffic: null` in that line
EAGER2="$EAGER --dtype bf16 --clips 4"
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_cfetch2 -o t -- $PY $EAGER2 > $OUT/conv_traffic_fetch_bf16.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_cwrite2 -o t -- $PY $EAGER2 > $OUT/conv_traffic_write_bf16.log 2>&1
$PY tools/summarize_prof.py convtraffic $OUT/pmc_cfetch2 $OUT/pmc_cwrite2 $OUT/traffic.json "conv_split:1000x600,interval=10,bf16,clips=4" > $OUT/conv_traffic_summary_bf16.log 2>&1
rm -rf $OUT/pmc_cfetch2 $OUT/pmc_cwrite2

# 5. per-op kernels (GPU-side durations) and the warp A/B
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ops -o t -- $PY tools/bench_ops.py > $OUT/bench_ops_microbench.txt 2>&1
cp $(find $OUT/trace_ops -name "*kernel_stats.csv" | head -1) $OUT/bench_ops_kernel_stats.csv 2>/dev/null
rm -rf $OUT/trace_ops
timeout 200 $PY tools/key_sections.py > $OUT/key_sections.txt 2>&1
LSFA_CONV_PIECES=3 timeout 200 $PY tools/key_sections.py > $OUT/key_sections_three_bf16_pieces.txt 2>&1
timeout 400 $PY tools/lab/key_batch_probe.py 1,2,3,4,6,8,12 2>&1 | tail -7 > $OUT/key_batch_probe.txt
timeout 300 $PY tools/lab/cur_batch_probe.py 2>&1 | tail -5 > $OUT/cur_batch_probe.txt
timeout 200 $PY tools/lab/stem_probe.py 2>&1 | tail -4 > $OUT/stem_probe.txt

# 6. the convolution kernels: every launch plan of the ring kernel per network shape against the library GEMM (error against
#    float64 + time, hipGraph-timed); per-kernel durations and counters of a few plans; the detection post-processing phase by phase
#    (SKIP_LAB=1: keep the committed lab tables - ~25 minutes of GPU time - when the convolution kernels did not change)
if [ "${SKIP_LAB:-0}" != "1" ]; then
timeout 1500 $PY tools/lab/conv_ring_lab.py --pieces 2,3,1 > $OUT/conv_ring_lab.txt 2>&1
timeout 900 $PY tools/lab/conv_ring_lab.py --quick --pieces 2 --batch 3 --shapes "res4,res5,res3,res2,feat" > $OUT/conv_ring_lab_batch3.txt 2>&1
fi
bash tools/lab/pmc_probe.sh > /dev/null 2> $OUT/pmc_probe.err; mv gpurun_out/pmc_probe_summary.txt gpurun_out/trace_probe_rows_split.txt $OUT/ 2>/dev/null
# 7. r3: kernel sequences of FlowNet and of one non-key frame (eager), multi-process determinism table
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/fn -o t -- $PY tools/backbone_only.py 20 flownet > /dev/null 2>&1
timeout 60 $PY tools/kernel_sequence.py $OUT/fn 20 > $OUT/flownet_kernel_sequence.txt 2>&1; rm -rf $OUT/fn
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/cf -o t -- $PY tools/curframe_only.py 30 > /dev/null 2>&1
timeout 60 $PY tools/kernel_sequence.py $OUT/cf 30 > $OUT/curframe_kernel_sequence.txt 2>&1; rm -rf $OUT/cf
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/cf9 -o t -- $PY tools/curframe_only.py 12 9 > /dev/null 2>&1
timeout 60 $PY tools/kernel_sequence.py $OUT/cf9 12 > $OUT/segment9_kernel_sequence.txt 2>&1; rm -rf $OUT/cf9
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/bb -o t -- $PY tools/backbone_only.py 12 backbone > /dev/null 2>&1
timeout 60 $PY tools/kernel_sequence.py $OUT/bb 12 --by-name > $OUT/backbone_kernels_by_name.txt 2>&1; rm -rf $OUT/bb
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/bb6 -o t -- $PY tools/backbone_only.py 8 backbone 6 > /dev/null 2>&1
timeout 60 $PY tools/kernel_sequence.py $OUT/bb6 8 --by-name > $OUT/backbone6_kernels_by_name.txt 2>&1
timeout 60 $PY tools/kernel_sequence.py $OUT/bb6 8 > $OUT/backbone6_kernel_sequence.txt 2>&1; rm -rf $OUT/bb6
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/bb12 -o t -- $PY tools/backbone_only.py 6 backbone 12 > /dev/null 2>&1     # r6: bench.py's default key group
timeout 60 $PY tools/kernel_sequence.py $OUT/bb12 6 --by-name > $OUT/backbone12_kernels_by_name.txt 2>&1; rm -rf $OUT/bb12
# 8. r4: where the pipelined loop's wall time goes per hardware queue
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/tl -o t -- $PY bench.py --steps 30 --no-cpu-baseline --no-parity --no-spread --no-frame-by-frame > /dev/null 2>&1
timeout 120 $PY tools/pipeline_timeline.py $OUT/tl 0.5 --chart 60 250 > $OUT/pipeline_timeline.txt 2>&1; rm -rf $OUT/tl
# 9. r6: ... and the schedule WITHOUT a tracer attached (the traced loop runs its queues one after the other: LABNOTES "Round 6", 13)
timeout 300 $PY tools/lab/schedule_probe.py --steps 40 2>&1 | grep -A200 "schedule of" > $OUT/schedule_probe.txt
ls -la $OUT
