set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4; mkdir -p $OUT; PY=python3
cp profiles/traffic.json $OUT/traffic.json
EAGER="bench.py --steps 6 --warmup 3 --settle-s 0 --no-graph --no-cpu-baseline --no-parity --no-spread"
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_cfetch -o t -- $PY $EAGER > $OUT/conv_traffic_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_cwrite -o t -- $PY $EAGER > $OUT/conv_traffic_write.log 2>&1
$PY tools/summarize_prof.py convtraffic $OUT/pmc_cfetch $OUT/pmc_cwrite $OUT/traffic.json "conv_split:1000x600,interval=10,f32" > $OUT/conv_traffic_summary.log 2>&1
rm -rf $OUT/pmc_cfetch $OUT/pmc_cwrite
tail -40 $OUT/conv_traffic_summary.log
