#!/usr/bin/env python
"""HBM traffic of the streaming hand-written kernels, per launch, for bench.py's `roofline.traffic`.

On the GPU box, three runs of the same launches (rocprofv3 cannot hold FETCH_SIZE and WRITE_SIZE in one pass,
MI355X_MICROARCH.md "rocprofv3 PMC slots"):

    rocprofv3 --pmc FETCH_SIZE  -d gpurun_out/pmc_fetch -o t --output-format csv -- python3 tools/traffic_probe.py run
    rocprofv3 --pmc WRITE_SIZE  -d gpurun_out/pmc_write -o t --output-format csv -- python3 tools/traffic_probe.py run
    rocprofv3 --kernel-trace --stats -d gpurun_out/pmc_trace -o t --output-format csv -- python3 tools/traffic_probe.py run
    python3 tools/traffic_probe.py summarize gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_trace gpurun_out/traffic.json

`summarize` writes, per shape key (the one bench.py looks up), the mean FETCH_SIZE / WRITE_SIZE (KiB as
rocprofv3 reports them), hbm_bytes_per_launch = 2 x FETCH + WRITE (the guide's gfx950 correction: 128-byte
read requests are tallied as 64 B; calibrated for wide coalesced reads — this kernel's tap gathers are 8-byte
accesses, so the read half is an upper-bound estimate), the algorithmic bytes and the mean duration from
the kernel trace.  Copy the result to profiles/traffic.json.
"""
import collections
import csv
import glob
import json
import os
import sys

C, H, W = 1024, 38, 63
ITERS = 10
# (key, maps per launch, epilogue): launches happen in this order, ITERS times each
PLAN = [("warp_bilinear:N=1,C=1024,H=38,W=63", 1, "cur"), ("warp_bilinear:N=1,C=1024,H=38,W=63|key", 1, "key"),
        ("warp_bilinear:N=32,C=1024,H=38,W=63", 32, "key"), ("warp_bilinear:N=32,C=1024,H=38,W=63|cur", 32, "cur"),
        ("aggregate_softmax2:N=1,C=1024,H=38,W=63", 1, "agg"), ("aggregate_softmax2:N=32,C=1024,H=38,W=63", 32, "agg")]


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from lsfa_amd import hip
    dev = 'cuda:0'
    g = torch.Generator(device=dev).manual_seed(0)
    for key, N, kind in PLAN:
        feat = torch.randn((N, C, H, W), device=dev, generator=g)
        other = torch.randn((N, C, H, W), device=dev, generator=g)
        out = torch.empty_like(feat)
        flow = torch.randn((N, 2, H, W), device=dev, generator=g) * 0.05 + torch.tensor([1.7, -0.6], device=dev).view(1, 2, 1, 1)
        res = torch.randn((N, 3, H, W), device=dev, generator=g)
        rw, rb = torch.randn((C, 3), device=dev, generator=g) * 0.01, torch.randn(C, device=dev, generator=g) * 0.01
        logits = torch.randn((2 * N, 1, H, W), device=dev, generator=g)
        torch.cuda.synchronize()
        for _ in range(ITERS):
            if kind == "key":
                hip.warp_bilinear(feat, flow, mul=other, out=out)
            elif kind == "cur":
                hip.warp_bilinear(feat, flow, add=other, res=res, res_w=rw, res_b=rb, out=out)
            else:
                hip.aggregate_softmax2(feat, other, logits, out=out)
        torch.cuda.synchronize()
        del feat, other, out
    print("traffic_probe: done")


def _kernel_rows(d, suffix):
    f = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not f:
        raise SystemExit("no %s under %s" % (suffix, d))
    rows = list(csv.DictReader(open(f[0])))
    return [r for r in rows if "warp_kernel" in r["Kernel_Name"] or "warp_staged_kernel" in r["Kernel_Name"] or "combine_kernel" in r["Kernel_Name"]]


def summarize(fetch_dir, write_dir, trace_dir, out_path):
    def counter(d, name):
        rows = [r for r in _kernel_rows(d, "counter_collection.csv") if r["Counter_Name"] == name]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        return [float(r["Counter_Value"]) for r in rows]
    fetch, write = counter(fetch_dir, "FETCH_SIZE"), counter(write_dir, "WRITE_SIZE")
    trace = _kernel_rows(trace_dir, "kernel_trace.csv")
    trace.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in trace]
    names = [r["Kernel_Name"] for r in trace]
    assert len(fetch) == len(write) == len(dur) == ITERS * len(PLAN), (len(fetch), len(write), len(dur))
    result = collections.OrderedDict()
    for i, (key, N, kind) in enumerate(PLAN):
        sl = slice(i * ITERS + 2, (i + 1) * ITERS)          # skip the first two launches of each shape
        mean = lambda v: sum(v[sl]) / len(v[sl])
        f_kib, w_kib = mean(fetch), mean(write)
        alg = N * (3 * C * H * W + 2 * H * W) * 4
        result[key] = {"kernel": names[i * ITERS + 2][:110], "fetch_size_kib_raw": round(f_kib, 1), "write_size_kib": round(w_kib, 1),
                       "hbm_bytes_per_launch": int((2 * f_kib + w_kib) * 1024), "algorithmic_bytes_per_launch": alg,
                       "traffic_over_algorithmic": round((2 * f_kib + w_kib) * 1024 / alg, 3), "avg_us_kernel_trace": round(mean(dur), 2),
                       "GBps_algorithmic": round(alg / mean(dur) / 1e3, 1)}
    with open(out_path, "w") as f:
        json.dump(result, f, indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "summarize":
        summarize(*sys.argv[2:6])
    else:
        run()
