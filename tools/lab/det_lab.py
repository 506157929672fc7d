#!/usr/bin/env python
"""Where the time goes inside det_class_kernel: phase boundaries of one class's workgroup.

    python tools/lab/det_lab.py [--iters 20]

Builds tools/lab/_build/libdet_lab.so = detpost.hip compiled with -DLSFA_DET_STAMPS (+ runtime.hip); the
kernel then leaves wall_clock64() stamps (100 MHz) at its phase boundaries for class 1.  Two inputs: every
roi passes the score threshold for every class (the bench's random-weight network) and a sparse case.
"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

PHASES = ["threshold+compaction+decode", "rank+relayout", "mask", "sweep", "write-out"]


def build():
    out = os.path.join(HERE, "_build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libdet_lab.so")
    csrc = os.path.join(ROOT, "lsfa_amd", "csrc")
    srcs = [os.path.join(csrc, "detpost.hip"), os.path.join(csrc, "runtime.hip")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in srcs):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-ffp-contract=off", "-fno-fast-math", "-DLSFA_DET_STAMPS", "-I",
                               os.path.join(ROOT, "include"), "-I", csrc] + srcs + ["-o", so])
    return ctypes.CDLL(so)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--build-only", action="store_true")
    args = ap.parse_args()
    lab = build()
    if args.build_only:
        return
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    R, ncls, H, W = 300, 31, 38, 63
    x1 = rs.uniform(0, W * 16 - 200, R); y1 = rs.uniform(0, H * 16 - 200, R)
    rois = np.stack([np.zeros(R), x1, y1, x1 + rs.uniform(30, 190, R), y1 + rs.uniform(30, 190, R)], 1).astype(np.float32)
    deltas = (0.15 * rs.randn(R, 8)).astype(np.float32)
    logits = (2 * rs.randn(R, ncls)).astype(np.float32)
    e = np.exp(logits - logits.max(1, keepdims=True))
    probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
    sparse = probs.copy(); sparse[:, 1:] *= 1e-3; sparse[:40, 1] = 0.5
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rois_d, deltas_d = t(rois), t(deltas)
    dets = torch.zeros((ncls, R, 5), dtype=torch.float64, device=dev)
    counts = torch.zeros(ncls, dtype=torch.int32, device=dev)
    keep = torch.zeros((ncls, R), dtype=torch.int32, device=dev)
    vp = lambda x: ctypes.c_void_p(x.data_ptr())
    for tag, p in (("all pass", probs), ("sparse", sparse)):
        p_d = t(p)
        acc = np.zeros(5)
        fine = np.zeros(4)
        for it in range(args.iters + 3):
            rc = lab.lsfa_det_postprocess(vp(rois_d), vp(deltas_d), vp(p_d), R, ncls, 8, 1, ctypes.c_double(H * 16.0),
                                          ctypes.c_double(W * 16.0), ctypes.c_double(1.0), ctypes.c_double(1e-3),
                                          ctypes.c_double(0.3), 300, vp(dets), vp(counts), vp(keep), None,
                                          ctypes.c_size_t(0), None)
            assert rc == 0
            torch.cuda.synchronize()
            st = (ctypes.c_longlong * 16)()
            assert lab.lsfa_det_lab_stamps(st) == 0
            if it >= 3:
                acc += np.diff(np.array(st[:6], dtype=np.float64)) / 100.0
                fine += np.diff(np.array(st[8:13], dtype=np.float64)) / 100.0
        acc /= args.iters
        print("%s: class 1 keeps %d of %d" % (tag, int(counts[1]), int((p[:, 1] > 1e-3).sum())))
        for name, us in zip(PHASES, acc):
            print("  %-30s %7.2f us" % (name, us))
        print("  %-30s %7.2f us" % ("total", acc.sum()))
        print("  sweep, block 0: loads %.2f, resolve %.2f, kept list %.2f, row OR %.2f us (each stamp costs ~0.3-1 us itself)"
              % tuple(fine / args.iters))


if __name__ == "__main__":
    main()
