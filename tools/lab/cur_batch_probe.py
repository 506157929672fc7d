#!/usr/bin/env python
"""Lab: what a non-key frame costs when the B non-key frames of a segment (same key feature) go through the network in one pass, batch axis
= frames (the reference's own batch test symbol does this: get_batch_test_symbol, resnet_v1_101_flownet_rfcn.py:661-751).  Each variant
captured as one hipGraph (network + lsfa_det_postprocess_batch) and replayed alone.  Argument: the batch sizes, e.g. 9,18,27."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lsfa_amd import hip
from lsfa_amd.config.config import lsfa_test_config
from lsfa_amd.symbols import params as P
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
dev = 'cuda:0'
H, W = 600, 1000
cfg = lsfa_test_config(key_frame_interval=10)
arg, aux = P.init_params(cfg, seed=0)
DT = torch.bfloat16 if os.environ.get('LSFA_PROBE_DTYPE') == 'bf16' else torch.float32      # bf16: the one-product mode (BASELINE configs[2])
cur = resnet_v1_101_flownet_rfcn(cfg).get_cur_test_symbol(cfg).bind(arg, aux, dev, dtype=DT)
R, ncls = cfg.TEST.RPN_POST_NMS_TOP_N, cfg.dataset.NUM_CLASSES


def graph_time(fn, n=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), torch.no_grad():
        fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


feat = torch.randn(1, 1024, 38, 63, device=dev)
for B in [int(v) for v in sys.argv[1].split(',')] if len(sys.argv) > 1 else (1, 2, 3, 5, 9):
    data = torch.rand(B, 3, H, W, device=dev) * 255
    im_info = torch.tensor([[H, W, 1.0]] * B, device=dev)
    mv = torch.randn(B, 2, 38, 63, device=dev) * 0.5
    res = torch.randn(B, 3, 38, 63, device=dev)
    bufs = (torch.zeros((B, ncls, R, 5), dtype=torch.float64, device=dev), torch.zeros((B, ncls), dtype=torch.int32, device=dev),
            torch.full((B, ncls, R), -1, dtype=torch.int32, device=dev))      # what core/graphs.py hands lsfa_det_postprocess_batch

    def frame():
        out = cur.forward(data=data, im_info=im_info, feat_key=feat, motion_vector=mv, res_diff=res)
        hip.det_postprocess_batch(out['rois_output'], out['bbox_pred_reshape_output'].reshape(B * R, -1), out['cls_prob_reshape_output'].reshape(B * R, -1),
                                  B, H, W, 1.0, bufs, nms_thresh=cfg.TEST.NMS, max_per_image=cfg.TEST.max_per_image, class_agnostic=cfg.CLASS_AGNOSTIC)
        return out

    def small():
        return cur.small_net_feature(data)
    t, ts = graph_time(frame), graph_time(small)
    print('frames per pass %d: non-key pass %8.1f us (%6.1f per frame)   of which small net %7.1f us (%6.1f per frame)' % (B, t, t / B, ts, ts / B), flush=True)
cur.check_status()
