#!/usr/bin/env python
"""Ring-kernel lab (r4): the network's convolution / GEMM shapes under every launch plan the ring kernel offers.

    python tools/lab/conv_ring_lab.py [--quick] [--pieces 2,3,1] > gpurun_out/conv_ring_lab.txt

Per shape and operand form: the launch plan's own choice (time, error against a float64 convolution), then the sweep
tile width nt x ring depth st x K slices through lsfa_conv_plan_override, and the tuned / default library GEMM for the 1x1
shapes.  Time = HIP events around `reps` back-to-back calls that cycle over 6 copies of the weights and 3 of the input
(in the frame path a layer's weights were last touched a frame ago: they come from the Infinity Cache, not from L2),
captured into one hipGraph per measurement (kernel time incl. launch gaps, not host time); median of 3 replays; the reduce pass
of a K-sliced plan is included.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lsfa_amd import hip  # noqa: E402

DEV = "cuda:0"
# name: (H, W, Cin, Cout, k, stride, dil, epilogue) ; epilogue: 'relu' | 'res2' (residual + second output) | 'nchw'
SHAPES = {
    "res4 conv1 1024->256": (38, 63, 1024, 256, 1, 1, 1, 'relu'),
    "res4 conv2 256->256": (38, 63, 256, 256, 3, 1, 1, 'relu'),
    "res4 conv3 256->1024": (38, 63, 256, 1024, 1, 1, 1, 'res2'),
    "res5 conv1 2048->512": (38, 63, 2048, 512, 1, 1, 1, 'relu'),
    "res5 conv3 512->2048": (38, 63, 512, 2048, 1, 1, 1, 'res2'),
    "res5 dcn 4608->512": (38, 63, 4608, 512, 1, 1, 1, 'relu'),
    "res5 sc 1024->2048": (38, 63, 1024, 2048, 1, 1, 1, 'none'),
    "res3 conv1 512->128": (75, 125, 512, 128, 1, 1, 1, 'relu'),
    "res3 conv2 128->128": (75, 125, 128, 128, 3, 1, 1, 'relu'),
    "res3 conv3 128->512": (75, 125, 128, 512, 1, 1, 1, 'res2'),
    "res2 conv1 256->64": (150, 250, 256, 64, 1, 1, 1, 'relu'),
    "res2 conv2 64->64": (150, 250, 64, 64, 3, 1, 1, 'relu'),
    "res2 conv3 64->256": (150, 250, 64, 256, 1, 1, 1, 'res2'),
    "feat 3x3 d6 2048->1024": (38, 63, 2048, 1024, 3, 1, 6, 'nchw'),
    "fuse 3x3 256->1024": (38, 63, 256, 1024, 3, 1, 1, 'nchw'),
    "rfcn 512->1920": (38, 63, 512, 1920, 1, 1, 1, 'none'),
    "rpn 512->64": (38, 63, 512, 64, 1, 1, 1, 'none'),
    "small 3x3 64->64": (38, 63, 64, 64, 3, 1, 1, 'relu'),
    "small 1x1 64->256": (38, 63, 64, 256, 1, 1, 1, 'res2'),
    # FlowNet-S on the half-resolution pair (resnet_v1_101_flownet_rfcn.py:153-169); the lab's ReLU stands in for its LeakyReLU
    "flow conv2 5x5/2 64->128": (150, 250, 64, 128, 5, 2, 1, 'relu'),
    "flow conv3 5x5/2 128->256": (75, 125, 128, 256, 5, 2, 1, 'relu'),
    "flow conv3_1 256->256": (38, 63, 256, 256, 3, 1, 1, 'relu'),
    "flow conv4 3x3/2 256->512": (38, 63, 256, 512, 3, 2, 1, 'relu'),
    "flow conv4_1 512->512": (19, 32, 512, 512, 3, 1, 1, 'relu'),
    "flow conv5 3x3/2 512->512": (19, 32, 512, 512, 3, 2, 1, 'relu'),
    "flow conv5_1 512->512": (10, 16, 512, 512, 3, 1, 1, 'relu'),
    "flow conv6 3x3/2 512->1024": (10, 16, 512, 1024, 3, 2, 1, 'relu'),
    "flow conv6_1 1024->1024": (5, 8, 1024, 1024, 3, 1, 1, 'relu'),
}


def timed(fn, reps, rounds=3):
    """us per call: `reps` calls captured into ONE hipGraph (issued eagerly the host's ~15 us per call would be the measurement for
    every launch shorter than that), replayed `rounds` times; median"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(0)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(reps):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / reps)
    del g
    return float(np.median(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--pieces', default='2,3')
    ap.add_argument('--shapes', default='')
    ap.add_argument('--reps', type=int, default=24)
    ap.add_argument('--no-sweep', action='store_true')
    ap.add_argument('--force', default='', help='kernel,nt,st,slices: time this plan alone (1 = mixed-role waves, 2 = loader / consumer waves)')
    ap.add_argument('--batch', type=int, default=1, help='images per call (the batched pipeline: 3 key fronts, 9 non-key frames)')
    args = ap.parse_args()
    pieces_list = [int(p) for p in args.pieces.split(',')]
    g = torch.Generator(device=DEV).manual_seed(0)
    names = [n for n in SHAPES if not args.shapes or any(s in n for s in args.shapes.split(','))]
    print("# %s   reps %d   torch %s" % (time.strftime('%Y-%m-%d %H:%M'), args.reps, torch.__version__))
    for name in names:
        H, W, ci, co, k, stride, dil, epi = SHAPES[name]
        pad = dil * (k // 2)
        NB = args.batch
        Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
        xs = [torch.relu(torch.randn((NB, H, W, ci), device=DEV, generator=g)) * 2.0 for _ in range(3)]
        ws = [torch.randn((co, ci, k, k), device=DEV, generator=g) * (1.0 / (ci * k * k) ** 0.5) for _ in range(6)]
        b = torch.randn(co, device=DEV, generator=g)
        res = torch.randn((NB, Ho, Wo, co), device=DEV, generator=g)
        sc2, sh2 = torch.rand(co, device=DEV, generator=g) + 0.5, torch.randn(co, device=DEV, generator=g)
        gf = 2.0 * NB * Ho * Wo * ci * co * k * k / 1e9
        ref = torch.nn.functional.conv2d(xs[0].permute(0, 3, 1, 2).double().cpu(), ws[0].double().cpu(), b.double().cpu(), stride=stride, padding=pad,
                                         dilation=dil)
        if epi == 'res2':
            ref = ref + res.permute(0, 3, 1, 2).double().cpu()
        if epi in ('relu', 'nchw'):
            ref = torch.relu(ref)
        scale = float(ref.abs().max())
        print("\n## %s   P=%d K=%d N=%d   %.2f GFLOP" % (name, NB * Ho * Wo, ci * k * k, co, gf))
        if args.force:
            pass
        elif k == 1 and stride == 1:      # the library GEMM of the same shape (untuned; bias / activation not included)
            X = [x.view(NB * H * W, ci) for x in xs]
            Wt = [w.view(co, ci).t().contiguous() for w in ws]
            t_lib = timed(lambda i: torch.mm(X[i % 3], Wt[i % 6]), args.reps)
            print("   library torch.mm (fp32 MFMA)                      %7.1f us   %6.1f TFLOP/s" % (t_lib, gf / t_lib * 1e3))
        for pieces in pieces_list:
            sws = [hip.SplitWeight(w, pieces=pieces) for w in ws]
            ams = [hip.amax_partial(x) for x in xs]
            slots = hip.amax_slots(1, DEV)[0]
            out = torch.empty((NB, co, Ho, Wo) if epi == 'nchw' else (NB, Ho, Wo, co), device=DEV)
            out2 = torch.empty_like(out)

            def call(i, sws=sws, ams=ams, out=out, out2=out2):
                kw = dict(amax_in=ams[i % 3], amax_out=slots, out=out)
                if epi == 'res2':
                    kw.update(residual=res, out2=out2, scale2=sc2, shift2=sh2)
                return hip.conv_split(xs[i % 3], sws[i % 6], b, stride, pad, dil, relu=epi in ('relu', 'nchw'), nchw=epi == 'nchw', **kw)
            if args.force:
                kn, nt, st, sl = [int(v) for v in args.force.split(',')]
                hip.conv_plan_override(kernel=kn, nt=nt, st=st, slices=sl)
                call(0)
                timed(call, args.reps, rounds=1)
                tt = timed(call, args.reps)
                print("   pieces %d   forced %s nt%d st%d s%d   %7.1f us   %6.1f TFLOP/s" % (pieces, {1: 'mix', 2: 'split', 4: 'mix8'}[kn], nt, st, sl, tt, gf / tt * 1e3))
                hip.conv_plan_override()
                continue
            hip.conv_plan_override()
            y = call(0)
            y = y[0] if isinstance(y, tuple) else y
            yc = (y if epi == 'nchw' else y.permute(0, 3, 1, 2)).double().cpu()
            err = float((yc - ref).abs().max()) / scale
            timed(call, args.reps, rounds=1)        # the first timed graph of a shape runs 10-15 % slow (clocks, caches): not the plan's fault
            t_auto = timed(call, args.reps)
            print("   pieces %d   plan's choice                            %7.1f us   %6.1f TFLOP/s   err/max %.2e" % (pieces, t_auto, gf / t_auto * 1e3, err))
            if args.no_sweep:
                continue
            rows = []
            for kern in (1, 2, 4):          # 1: mixed-role waves, 2: loader / consumer waves, 4: eight mixed-role waves on 256-pixel tiles
                for nt in (2, 4):
                    if co % (32 * nt) or (kern == 4 and (nt != 4 or pieces == 3)):
                        continue
                    for st in (2, 3, 4):
                        if (nt == 4 and pieces == 3 and st == 4) or (kern == 4 and st == 4):
                            continue
                        for s in ((1, 2, 3, 4, 6, 8, 12) if not args.quick else (1, 3, 6)):
                            if s > 1 and (ci * k * k // 32) // s < 3:
                                continue
                            try:
                                hip.conv_plan_override(kernel=kern, nt=nt, st=st, slices=s)
                                call(0)
                                rows.append((timed(call, max(8, args.reps // 2), rounds=2), kern, nt, st, s))
                            except hip.LsfaError:
                                pass
            hip.conv_plan_override()
            rows.sort()
            tag = {1: "mix", 2: "split", 4: "mix8"}
            print("      best plans (ring kernel forced): " + "   ".join("%s nt%d st%d s%d %.1f" % (tag[kn], nt, st, s, tt) for tt, kn, nt, st, s in rows[:6]))
            by = {}
            for tt, kn, nt, st, s in rows:
                by.setdefault((kn, nt, st), []).append((s, tt))
            for (kn, nt, st), v in sorted(by.items()):
                print("      %-5s nt%d st%d: " % (tag[kn], nt, st) + "  ".join("s%d %.1f" % (s, tt) for s, tt in sorted(v)))


if __name__ == '__main__':
    main()
