#!/usr/bin/env python
"""Per-op microbenchmark of the HIP entry points at BASELINE shapes (1000x600 -> 38x63x1024).

Each op is launched `iters` times back to back on one stream between two events; the
reported time therefore includes the ~1.5 us inter-kernel boundary.  Algorithmic bytes follow
SURVEY.md §8(d)."""
import argparse
import json
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lsfa_amd import hip


def timeit(fn, iters=50, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def rpn_inputs(rs, H, W, A=9, frac=0.02, trained=False):
    logit = rs.randn(1, 2, A * H, W).astype(np.float32) * 2
    logit[:, 1] += np.log(frac / (1 - frac))
    e = np.exp(logit - logit.max(1, keepdims=True))
    prob = (e / e.sum(1, keepdims=True)).astype(np.float32).reshape(1, 2 * A, H, W)
    std = 0.05 if trained else 1.0
    deltas = (rs.randn(1, 4 * A, H, W) * std * np.tile([0.1, 0.1, 0.4, 0.4], A)[None, :, None, None]).astype(np.float32)
    if trained:   # a few objects: anchors near them score high and regress to the same box -> heavy overlap
        prob4 = prob.reshape(1, 2, A, H, W)
        for _ in range(6):
            cy, cx = rs.randint(4, H - 4), rs.randint(4, W - 4)
            prob4[0, 1, :, cy - 3:cy + 4, cx - 3:cx + 4] = rs.uniform(0.6, 0.999, (A, 7, 7))
        prob = prob4.reshape(1, 2 * A, H, W)
    return prob, deltas


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--H', type=int, default=38)
    ap.add_argument('--W', type=int, default=63)
    ap.add_argument('--proposal-plan', default='auto', choices=['auto', 'single', 'chip'])
    args = ap.parse_args()
    dev = 'cuda:0'
    hip.proposal_set_plan(args.proposal_plan)
    H, W, C = args.H, args.W, 1024
    HW = H * W
    rs = np.random.RandomState(0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    out = {}
    feat, feat2 = torch.randn(1, C, H, W, device=dev), torch.randn(1, C, H, W, device=dev)
    flow = (torch.randn(1, 2, H, W, device=dev) * 0.3 + 1.5)
    res, res_w, res_b = torch.randn(1, 3, H, W, device=dev), torch.randn(C, 3, device=dev) * 0.01, torch.randn(C, device=dev) * 0.01
    o = torch.empty_like(feat)
    b3 = (3 * C * HW + 2 * HW) * 4
    b2 = (2 * C * HW + 2 * HW) * 4
    LG = torch.randn(2, 1, H, W, device=dev)
    for name, fn, nbytes in (
            ('warp_plain', lambda: hip.warp_bilinear(feat, flow, out=o), b2),
            ('warp_key(x scale)', lambda: hip.warp_bilinear(feat, flow, mul=feat2, out=o), b3),
            ('warp_cur(+res+add)', lambda: hip.warp_bilinear(feat, flow, add=feat2, res=res, res_w=res_w, res_b=res_b, out=o), b3),
            ('aggregate_softmax2', lambda: hip.aggregate_softmax2(feat, feat2, LG, out=o), b3),
            # streaming references of the same size: what a kernel moving these bytes with no gathers costs
            ('ref: torch copy (1 map)', lambda: o.copy_(feat), b2),
            ('ref: torch add (2 maps -> 1)', lambda: torch.add(feat, feat2, out=o), b3)):
        us = timeit(fn, args.iters)
        out[name] = dict(us=round(us, 2), GBps=round(nbytes / us / 1e3, 1), bytes=nbytes)
    # Fgfa cosine weights at LSFA's shape: 2048-channel embeddings on the 38x63 map
    ew, ec = torch.randn(1, 2048, H, W, device=dev), torch.randn(1, 2048, H, W, device=dev)
    us = timeit(lambda: hip.aggregate_cosine(feat, feat2, ew, ec, out=o), args.iters)
    out['aggregate_cosine(E=2048)'] = dict(us=round(us, 2), bytes=2 * 2 * 2048 * HW * 4 + b3)
    # compressed-domain motion vectors, one 1000x600 P-frame of 16x16 macroblocks
    acc = hip.MotionVectorAccumulator(1000, 600, dev)
    blocks = [[-1, 16, 16, bx + 8 - rs.randint(-9, 10), by + 8 - rs.randint(-9, 10), bx + 8, by + 8]
              for by in range(0, 600, 16) for bx in range(0, 1000, 16)]
    mvs = torch.tensor(blocks, dtype=torch.int32, device=dev)
    us = timeit(lambda: acc.add_frame(mvs, max_block_area=256), args.iters)
    out['mv_accumulate(1000x600)'] = dict(us=round(us, 2), bytes=1000 * 600 * (4 + 4 + 8 + 8))
    # own fp32 MFMA convolution: stage-3 conv2
    xr = torch.randn(1, H, W, 256, device=dev)
    wk = hip.conv_weight_kc(torch.randn(256, 256, 3, 3, device=dev) * 0.02)
    bb = torch.randn(256, device=dev)
    us = timeit(lambda: hip.conv_nhwc(xr, wk, bb, 3, 3, 1, 1, 1, relu=True), args.iters)
    out['conv_nhwc (fp32 MFMA) 3x3 256->256 @38x63'] = dict(us=round(us, 2), TFLOPs=round(2 * HW * 256 * 2304 / us / 1e6, 1))
    # the same and the two big ones on the split-bf16 kernel
    sw = hip.SplitWeight(torch.randn(256, 256, 3, 3, device=dev) * 0.02)
    us = timeit(lambda: hip.conv_split(xr, sw, bb, 1, 1, 1, relu=True), args.iters)
    out['conv_split 3x3 256->256 @38x63 (res4 conv2)'] = dict(us=round(us, 2), TFLOPs=round(2 * HW * 256 * 2304 / us / 1e6, 1))
    sw = hip.SplitWeight(torch.randn(1024, 256, 3, 3, device=dev) * 0.02)
    b1k = torch.randn(1024, device=dev)
    us = timeit(lambda: hip.conv_split(xr, sw, b1k, 1, 1, 1, nchw=True), args.iters)
    out['conv_split 3x3 256->1024 @38x63 (fuse, NCHW out)'] = dict(us=round(us, 2), TFLOPs=round(2 * HW * 1024 * 2304 / us / 1e6, 1))
    x2k = torch.randn(1, H, W, 2048, device=dev)
    sw = hip.SplitWeight(torch.randn(1024, 2048, 3, 3, device=dev) * 0.01)
    us = timeit(lambda: hip.conv_split(x2k, sw, b1k, 1, 6, 6, relu=True, nchw=True), max(args.iters // 4, 3))
    out['conv_split 3x3 d6 2048->1024 @38x63 (feat_conv_3x3)'] = dict(us=round(us, 2), TFLOPs=round(2 * HW * 1024 * 18432 / us / 1e6, 1))
    del sw, x2k
    # ResNet stem: backbone (600x1000) and small net (frame / 4)
    img = torch.rand(1, 3, 600, 1000, device=dev) * 255
    w0 = hip.stem_weight_layout(torch.randn(64, 3, 7, 7, device=dev) * 0.05)
    b0, s3, t3 = torch.randn(64, device=dev), torch.rand(3, device=dev), torch.rand(3, device=dev)
    us = timeit(lambda: hip.avgpool_nchw(img, 4), args.iters)
    out['avgpool_nchw 4x4 (600x1000)'] = dict(us=round(us, 2), GBps=round(img.numel() * 4 * 17 / 16 / us / 1e3, 1))
    small = hip.avgpool_nchw(img, 4)
    for tag, xin in (('stem_conv7x7s2 (600x1000)', img), ('stem_conv7x7s2 (150x250)', small)):
        us = timeit(lambda: hip.stem_conv(xin, w0, b0, s3, t3), args.iters)
        ho, wo = (xin.shape[2] - 1) // 2 + 1, (xin.shape[3] - 1) // 2 + 1
        out[tag] = dict(us=round(us, 2), TFLOPs=round(2 * ho * wo * 64 * 147 / us / 1e6, 2))
        yy = hip.stem_conv(xin, w0, b0, s3, t3)
        us = timeit(lambda: hip.maxpool3x3s2_nhwc(yy), args.iters)
        out[tag.replace('stem_conv7x7s2', 'maxpool3x3s2_nhwc')] = dict(us=round(us, 2))
    # PSROI / head
    cls_map, box_map = torch.randn(1, 31 * 49, H, W, device=dev), torch.randn(1, 8 * 49, H, W, device=dev)
    rois_np = np.zeros((300, 5), np.float32)
    cx, cy = rs.uniform(0, W * 16, 300), rs.uniform(0, H * 16, 300)
    w_, h_ = rs.uniform(30, 400, 300), rs.uniform(30, 400, 300)
    rois_np[:, 1], rois_np[:, 2] = np.clip(cx - w_ / 2, 0, W * 16 - 1), np.clip(cy - h_ / 2, 0, H * 16 - 1)
    rois_np[:, 3], rois_np[:, 4] = np.clip(cx + w_ / 2, 0, W * 16 - 1), np.clip(cy + h_ / 2, 0, H * 16 - 1)
    rois = t(rois_np)
    hb = (1519 + 392) * HW * 4 + 300 * 39 * 4
    us = timeit(lambda: hip.rfcn_head(cls_map, box_map, rois), args.iters)
    out['rfcn_head_fused'] = dict(us=round(us, 2), GBps=round(hb / us / 1e3, 1), bytes=hb)
    ps = torch.randn(1, H, W, 49, 39, device=dev)
    us = timeit(lambda: hip.rfcn_head_ps(ps, rois, 31, 8), args.iters)
    out['rfcn_head_ps_layout'] = dict(us=round(us, 2), GBps=round(hb / us / 1e3, 1), bytes=hb)
    pb = 1519 * HW * 4 + 300 * 31 * 49 * 4
    us = timeit(lambda: hip.psroi_pool(cls_map, rois, 0.0625, 31, 7, 7), args.iters)
    out['psroi_pool_cls'] = dict(us=round(us, 2), GBps=round(pb / us / 1e3, 1), bytes=pb)
    # proposal: untrained-like (few overlaps) and trained-like (heavy overlaps)
    im_info = t(np.array([[H * 16 - 8, W * 16 - 8, 1.0]], np.float32))
    for tag, trained in (('proposal_sparse', False), ('proposal_clustered', True)):   # RPN-like inputs
        prob, deltas = rpn_inputs(rs, H, W, trained=trained)
        op = hip.ProposalOp(rpn_min_size=0)
        p_, d_ = t(prob), t(deltas)
        us = timeit(lambda: op(p_, d_, im_info), args.iters)
        out[tag] = dict(us=round(us, 2))
    # det post
    R, ncls = 300, 31
    deltas = t((0.15 * rs.randn(R, 8)).astype(np.float32))
    logits = (2 * rs.randn(R, ncls)).astype(np.float32)
    e = np.exp(logits - logits.max(1, keepdims=True))
    probs = t((e / e.sum(1, keepdims=True)).astype(np.float32))
    bufs = (torch.zeros((ncls, R, 5), dtype=torch.float64, device=dev), torch.zeros(ncls, dtype=torch.int32, device=dev),
            torch.zeros((ncls, R), dtype=torch.int32, device=dev))
    us = timeit(lambda: hip.det_postprocess(rois, deltas, probs, H * 16, W * 16, 1.0, out=bufs), args.iters)
    out['det_postprocess(all pass thresh)'] = dict(us=round(us, 2))
    sparse = probs.clone()
    sparse[:, 1:] *= 1e-3
    sparse[:40, 3] = 0.5
    us = timeit(lambda: hip.det_postprocess(rois, deltas, sparse, H * 16, W * 16, 1.0, out=bufs), args.iters)
    out['det_postprocess(sparse)'] = dict(us=round(us, 2))
    # dcn im2col + bn
    x = torch.randn(1, 512, H, W, device=dev)
    off = torch.randn(1, 72, H, W, device=dev)
    col = torch.empty(1, 512 * 9, HW, device=dev)
    us = timeit(lambda: hip.deform_im2col(x, off, 3, 3, 2, 1, 2, 4, out=col), args.iters)
    out['deform_im2col(512ch)'] = dict(us=round(us, 2), GBps=round((512 * HW * 4 * 10 + 72 * HW * 4) / us / 1e3, 1))
    xb = torch.randn(1, 1024, H, W, device=dev)
    sc, sh = torch.rand(1024, device=dev), torch.rand(1024, device=dev)
    us = timeit(lambda: hip.scale_shift_relu(xb, sc, sh, True, out=o), args.iters)
    out['scale_shift_relu(1024ch)'] = dict(us=round(us, 2), GBps=round(2 * C * HW * 4 / us / 1e3, 1))
    for k, v in out.items():
        print("%-36s %s" % (k, json.dumps(v)))


if __name__ == '__main__':
    main()


def batched():
    """HBM-bound regime: 32 feature maps per launch, > 256 MiB of traffic."""
    dev = 'cuda:0'
    N, C, H, W = 32, 1024, 38, 63
    HW = H * W
    feat, add = torch.randn(N, C, H, W, device=dev), torch.randn(N, C, H, W, device=dev)
    flow = torch.randn(N, 2, H, W, device=dev) * 0.3 + 1.5
    res, res_w, res_b = torch.randn(N, 3, H, W, device=dev), torch.randn(C, 3, device=dev) * 0.01, torch.randn(C, device=dev) * 0.01
    out = torch.empty_like(feat)
    b3 = N * (3 * C * HW + 2 * HW) * 4
    us = timeit(lambda: hip.warp_bilinear(feat, flow, add=add, res=res, res_w=res_w, res_b=res_b, out=out), 20, 3)
    print("warp_cur N=32      us %.1f  GB/s %.0f  bytes %d" % (us, b3 / us / 1e3, b3))
    us = timeit(lambda: hip.warp_bilinear(feat, flow, out=out), 20, 3)
    print("warp_plain N=32    us %.1f  GB/s %.0f" % (us, N * (2 * C * HW + 2 * HW) * 4 / us / 1e3))
    a2 = torch.randn(C * N, H, W, device=dev)
    us = timeit(lambda: out.copy_(feat), 20, 3)
    print("torch copy N=32    us %.1f  GB/s %.0f" % (us, 2 * N * C * HW * 4 / us / 1e3))
    sc, sh = torch.rand(C, device=dev), torch.rand(C, device=dev)
    us = timeit(lambda: hip.scale_shift_relu(feat, sc, sh, True, out=out), 20, 3)
    print("scale_shift N=32   us %.1f  GB/s %.0f" % (us, 2 * N * C * HW * 4 / us / 1e3))


if __name__ == '__main__' and os.environ.get('LSFA_BATCHED') == '1':
    batched()
