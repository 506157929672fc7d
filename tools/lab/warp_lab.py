#!/usr/bin/env python
"""A/B timing of the warp kernel variants on one device, interleaved rounds in one process.

    python tools/lab/warp_lab.py [--maps 1,32] [--rounds 15]

Builds tools/lab/_build/libwarp_lab.so (hipcc, gfx950), checks every variant bit-exact against
lsfa_warp_bilinear on the same inputs, then times them: per round each variant runs `iters` launches
back to back between two events on one stream; the table reports the median and minimum over rounds.
For GPU-side durations run it under `rocprofv3 --kernel-trace --stats` (the variants are different
template instances, so the stats list them separately).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from lsfa_amd import hip  # noqa: E402

VARIANTS = {0: "r1 kernel", 1: "round-2 rewrite (wave items, 8 ch, one batch, operands first), launch_bounds(256,1)",
            2: "generic: tiled grid, 8 ch, batches of 4 (= r1 structure)", 3: "generic: wave items, 8 ch, batches of 4",
            4: "generic: wave items, 8 ch, one batch of 8", 5: "generic: wave items, 8 ch, batches of 2",
            6: "generic: wave items, one batch of 8, operands hoisted", 7: "generic: tiled grid, one batch of 8",
            8: "generic: wave items, 4 ch/wave", 9: "generic: wave items, 16 ch/wave, batches of 4",
            10: "generic: wave items, batches of 4, taps as 4-byte loads", 11: "generic: wave items, 64-thread workgroups",
            12: "generic: wave items, 1024-thread workgroups", 13: "generic: wave items, 16 ch/wave, batches of 8",
            14: "generic: wave items, 4 ch/wave, batches of 2", 15: "round-2 rewrite, launch_bounds(256,2)",
            16: "round-2 rewrite + taps shared between the two pixels of a lane",
            17: "r3 LDS-staged planes (DMA ring): 640 thr, 3 slots, 8 ch/WG", 18: "r3 LDS-staged: 640 thr, 3 slots, 16 ch/WG",
            19: "r3 LDS-staged: 640 thr, 4 slots, 16 ch/WG", 20: "r3 LDS-staged: 320 thr, 3 slots, 16 ch/WG",
            21: "r3 LDS-staged: 320 thr, 4 slots, 16 ch/WG", 22: "r3 LDS-staged: 256 thr, 3 slots, 16 ch/WG",
            23: "r3 LDS-staged: 640 thr, 3 slots, 4 ch/WG", 24: "r3 LDS-staged: 640 thr, 3 slots, 32 ch/WG",
            25: "r3 LDS-staged: 640 thr, 3 slots, 2 ch/WG", 26: "r3 LDS-staged: 640 thr, 4 slots, 8 ch/WG",
            27: "r3 LDS-staged: 256 thr, 3 slots, 8 ch/WG", 28: "r3 LDS-staged: 256 thr, 3 slots, 4 ch/WG",
            29: "r3 LDS-staged: 1024 thr, 3 slots, 8 ch/WG", 30: "r3 LDS-staged: 1024 thr, 3 slots, 4 ch/WG",
            31: "r3 LDS-staged: 512 thr, 3 slots, 8 ch/WG", 32: "r3 LDS-staged: 512 thr, 3 slots, 4 ch/WG",
            99: "NOT a warp: out = feat (x|+) operand, 16-byte streams (the floor for these bytes)"}
NOT_A_WARP = {99}


def build():
    out = os.path.join(HERE, "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libwarp_lab.so")
    src = os.path.join(HERE, "warp_lab.hip")
    deps = [src] + [os.path.join(HERE, h) for h in ("warp_r1_kernel.h", "warp_variants.h", "warp_r2_attempt.h", "warp_lds.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-ffp-contract=off", "-fno-fast-math", "-I", os.path.join(ROOT, "include"), "-I",
                               os.path.join(ROOT, "lsfa_amd", "csrc"), "-I", HERE, src, "-o", so])
    return ctypes.CDLL(so)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--maps", default="1,32")
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--H", type=int, default=38)
    ap.add_argument("--W", type=int, default=63)
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--variants", default="", help="comma-separated subset (default: all)")
    args = ap.parse_args()
    lab = build()
    if args.build_only:
        return
    dev = "cuda:0"
    if args.variants:
        keep = [int(x) for x in args.variants.split(",")]
        for v in list(VARIANTS):
            if v not in keep:
                del VARIANTS[v]
    C, H, W = 1024, args.H, args.W
    vp, ci = ctypes.c_void_p, ctypes.c_int
    P = lambda t: vp(t.data_ptr()) if t is not None else None
    report = {}
    for N in [int(x) for x in args.maps.split(",")]:
        g = torch.Generator(device=dev).manual_seed(N)
        feat = torch.randn((N, C, H, W), device=dev, generator=g)
        other = torch.randn((N, C, H, W), device=dev, generator=g)
        # flow like the bench's: a global motion of a few cells plus small noise (cur path) / a smooth field (key path)
        flow = torch.randn((N, 2, H, W), device=dev, generator=g) * 0.05 + torch.tensor([1.7, -0.6], device=dev).view(1, 2, 1, 1)
        res = torch.randn((N, 3, H, W), device=dev, generator=g)
        res_w, res_b = torch.randn((C, 3), device=dev, generator=g) * 0.01, torch.randn(C, device=dev, generator=g) * 0.01
        out = torch.empty_like(feat)
        modes = {"key (x scale map)": dict(mul=other, add=None, res=None),
                 "cur (+res +small)": dict(mul=None, add=other, res=res)}
        nbytes = (3 * C * H * W + 2 * H * W) * 4 * N
        for mode, kw in modes.items():
            want = hip.warp_bilinear(feat, flow, mul=kw["mul"], add=kw["add"], res=kw["res"], res_w=res_w, res_b=res_b)

            def run(v):
                rc = lab.lab_warp(ci(v), P(feat), ci(N), P(flow), ci(N), ci(C), ci(H), ci(W), P(kw["mul"]), P(kw["add"]),
                                  P(kw["res"]), ci(3 if kw["res"] is not None else 0), P(res_w), P(res_b), P(out),
                                  vp(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, (v, rc)
            for v in VARIANTS:
                out.zero_()
                run(v)
                torch.cuda.synchronize()
                assert v in NOT_A_WARP or torch.equal(out, want), "variant %d differs from lsfa_warp_bilinear (%s, N=%d)" % (v, mode, N)
            times = {v: [] for v in VARIANTS}
            iters = args.iters if N == 1 else max(3, args.iters // 4)
            for _ in range(args.rounds):
                for v in VARIANTS:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(iters):
                        run(v)
                    b.record()
                    torch.cuda.synchronize()
                    times[v].append(a.elapsed_time(b) * 1e3 / iters)
            for v in VARIANTS:
                med, mn = float(np.median(times[v])), float(np.min(times[v]))
                report["N=%d %s v%d" % (N, mode, v)] = dict(variant=VARIANTS[v], median_us=round(med, 2), min_us=round(mn, 2),
                                                           GBps_at_median=round(nbytes / med / 1e3, 1))
                print("N=%-3d %-18s v%-2d %-66s median %8.2f us  min %8.2f us  %7.1f GB/s" % (
                    N, mode, v, VARIANTS[v], med, mn, nbytes / med / 1e3))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "warp_lab.json"), "w") as f:
        json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
