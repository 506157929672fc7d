#!/usr/bin/env python
"""Regenerate lsfa_amd/tuned/gemm_gfx950.csv: run the bench workload (fp32, then bf16) with TunableOp
tuning on and collect the chosen solutions.  Run on an MI355X:
    python tools/tune_gemms.py gpurun_out/gemm_gfx950.csv     # then copy into lsfa_amd/tuned/"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lsfa_amd import tuning
out = sys.argv[1]
if os.path.exists(out):
    os.remove(out)
tuning.enable(tune_missing=True, results_file=out)
torch.backends.cudnn.benchmark = True
import bench
for dtype in ('f32', 'bf16'):
    sys.argv = ['bench.py', '--no-cpu-baseline', '--dtype', dtype, '--steps', '1', '--warmup', '1', '--lanes', '1']
    args = bench.parse()
    r = bench.Runner(args, 0, 'cuda:0')
    r.prime()
    r.step(0)
    torch.cuda.synchronize()
    print(dtype, 'done', flush=True)
import torch.cuda.tunable as T
print(len(T.get_results()), 'tuned entries ->', out)
