#!/bin/bash
# PMC / kernel-trace probe of one convolution shape (tools/lab/conv_trace_probe.py); run from anywhere: works in the repository root
set -u
export TMPDIR=/tmp
cd "$(dirname "$0")/../.."
export PROBE_KERNEL=2
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS"; do
  tag=$(echo $P | cut -d' ' -f1)
  rocprofv3 --pmc $P --output-format csv -d gpurun_out/pmc_$tag -- python3 tools/lab/conv_trace_probe.py > gpurun_out/pmc_$tag.log 2>&1
done
python tools/lab/pmc_rows.py gpurun_out/pmc_* > gpurun_out/pmc_probe_summary.txt 2>&1
rm -rf gpurun_out/pmc_SQ_* gpurun_out/pmc_TCC_*
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_probe -- python3 tools/lab/conv_trace_probe.py > gpurun_out/trace_probe.log 2>&1; python tools/lab/trace_rows.py gpurun_out/trace_probe > gpurun_out/trace_probe_rows_split.txt 2>&1; rm -rf gpurun_out/trace_probe
wc -l gpurun_out/pmc_probe_summary.txt
