#!/usr/bin/env python
"""One non-key frame (small net + MV warp + heads + detection post-processing) eagerly, n times, for rocprofv3 kernel traces
(tools/kernel_sequence.py prints one steady-state repetition)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lsfa_amd import hip
from lsfa_amd.config.config import lsfa_test_config
from lsfa_amd.symbols import params as P
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn

dev = 'cuda:0'
cfg = lsfa_test_config(key_frame_interval=10)
arg, aux = P.init_params(cfg, seed=0)
cur = resnet_v1_101_flownet_rfcn(cfg).get_cur_test_symbol(cfg).bind(arg, aux, dev)
H, W = 600, 1000
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1            # frames per pass (the batched pipeline's segments: 9)
data = torch.rand(F, 3, H, W, device=dev) * 255
im_info = torch.tensor([[H, W, 1.0]] * F, device=dev)
feat = torch.randn(1, 1024, 38, 63, device=dev)
mv = torch.randn(F, 2, 38, 63, device=dev) * 0.5
res = torch.randn(F, 3, 38, 63, device=dev)
R, ncls = cfg.TEST.RPN_POST_NMS_TOP_N, cfg.dataset.NUM_CLASSES
bufs = (torch.zeros((F, ncls, R, 5), dtype=torch.float64, device=dev), torch.zeros((F, ncls), dtype=torch.int32, device=dev),
        torch.full((F, ncls, R), -1, dtype=torch.int32, device=dev))      # what core/graphs.py hands lsfa_det_postprocess_batch: static buffers
with torch.no_grad():
    for _ in range(n):
        out = cur.forward(data=data, im_info=im_info, feat_key=feat, motion_vector=mv, res_diff=res)
        hip.det_postprocess_batch(out['rois_output'], out['bbox_pred_reshape_output'].reshape(F * R, -1), out['cls_prob_reshape_output'].reshape(F * R, -1),
                                  F, H, W, 1.0, bufs, nms_thresh=cfg.TEST.NMS, max_per_image=cfg.TEST.max_per_image, class_agnostic=cfg.CLASS_AGNOSTIC)
torch.cuda.synchronize()
