#!/usr/bin/env python
"""The shader clock the matrix-heavy kernels actually run at: a one-wave probe kernel on its own stream counts shader cycles
(s_memtime) per tick of the constant 100 MHz counter (s_memrealtime) while, on another stream, nothing / feat_conv_3x3 / res4 conv2 /
the library's fp32 GEMM run back to back.  The roofline peaks are quoted at the nominal clock; this says what the part sustains."""
import ctypes
import os
import subprocess
import sys
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from lsfa_amd import hip  # noqa: E402


def build():
    out = os.path.join(HERE, "_build")
    os.makedirs(out, exist_ok=True)
    so, src = os.path.join(out, "libclock_probe.so"), os.path.join(HERE, "clock_probe.hip")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", src, "-o", so])
    return ctypes.CDLL(so)


def main():
    lab = build()
    if "--build-only" in sys.argv:
        return
    dev = "cuda:0"
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    probe_stream, work_stream = torch.cuda.Stream(), torch.cuda.Stream()

    def probe(ms=2.0):
        lab.clock_probe(ctypes.c_longlong(int(ms * 1e5)), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(probe_stream.cuda_stream))
        probe_stream.synchronize()
        c, r = out.tolist()
        return c / (r / 100.0) / 1e3          # GHz: cycles per microsecond / 1000

    g = torch.Generator(device=dev).manual_seed(0)
    NB = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 1
    if "--r5" in sys.argv:      # r5: the batched shapes under the ablation switches of LSFA_CONV_TILE_ORDER (bits 4 / 5: no copies / no arithmetic)
        xf = torch.relu(torch.randn((NB, 38, 63, 2048), device=dev, generator=g))
        wf = hip.SplitWeight(torch.randn((1024, 2048, 3, 3), device=dev, generator=g) * 0.01)
        w5 = hip.SplitWeight(torch.randn((512, 2048, 1, 1), device=dev, generator=g) * 0.02)
        amf = hip.amax_partial(xf)
        loads5 = {"idle": (lambda: None, 1),
                  "feat_conv_3x3 x%d" % NB: (lambda: hip.conv_split(xf, wf, None, 1, 6, 6, relu=True, amax_in=amf), 60),
                  "res5 conv1 x%d" % NB: (lambda: hip.conv_split(xf, w5, None, 1, 0, 1, relu=True, amax_in=amf), 600)}
        for name, (fn, n) in loads5.items():
            with torch.cuda.stream(work_stream):
                for _ in range(3):
                    fn()
            torch.cuda.synchronize()
            with torch.cuda.stream(work_stream):
                for _ in range(n):
                    fn()
            ghz = [probe(2.0) for _ in range(5)]
            busy = not work_stream.query()
            torch.cuda.synchronize()
            print("%-30s LSFA_CONV_TILE_ORDER=%-3s shader clock %.2f GHz (probes: %s)%s" % (name, os.environ.get("LSFA_CONV_TILE_ORDER", "0"), sorted(ghz)[len(ghz) // 2],
                  " ".join("%.2f" % v for v in ghz), "" if busy or name == "idle" else "   [work finished before the last probe]"), flush=True)
        return
    x_feat = torch.randn((1, 38, 63, 2048), device=dev, generator=g)
    w_feat = hip.SplitWeight(torch.randn((1024, 2048, 3, 3), device=dev, generator=g) * 0.01)
    x_res4 = torch.randn((1, 38, 63, 256), device=dev, generator=g)
    w_res4 = hip.SplitWeight(torch.randn((256, 256, 3, 3), device=dev, generator=g) * 0.05)
    a, b = torch.randn((2394, 1024), device=dev, generator=g), torch.randn((1024, 256), device=dev, generator=g)
    loads = {
        "idle": lambda: None,
        "feat_conv_3x3 (split-bf16, 128x128 tiles)": lambda: hip.conv_split(x_feat, w_feat, None, 1, 6, 6, relu=True),
        "res4 conv2 (split-bf16)": lambda: hip.conv_split(x_res4, w_res4, None, 1, 1, 1, relu=True),
        "library fp32 GEMM 2394x1024x256": lambda: torch.mm(a, b),
        "elementwise 1 GB (torch.mul)": None,
    }
    big = torch.randn((256 << 20) // 4, device=dev, generator=g)
    loads["elementwise 1 GB (torch.mul)"] = lambda: torch.mul(big, 1.5, out=big)
    for name, fn in loads.items():
        with torch.cuda.stream(work_stream):
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        # keep the work stream busy for ~60 ms, probe in the middle
        with torch.cuda.stream(work_stream):
            n = 1 if name == "idle" else (400 if name.startswith("feat") else 3000)
            for _ in range(n):
                fn()
        ghz = [probe(2.0) for _ in range(5)]
        busy = not work_stream.query()
        torch.cuda.synchronize()
        print("%-46s shader clock %.2f GHz (probes: %s)%s" % (name, sorted(ghz)[len(ghz) // 2], " ".join("%.2f" % v for v in ghz),
                                                               "" if busy or name == "idle" else "   [work finished before the last probe]"), flush=True)


if __name__ == "__main__":
    main()
