#!/usr/bin/env python
"""One non-key frame (small net + MV warp + heads + detection post-processing) eagerly, n times, for rocprofv3 kernel traces
(tools/kernel_sequence.py prints one steady-state repetition)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from lsfa_amd import hip, tuning
from lsfa_amd.config.config import lsfa_test_config
from lsfa_amd.symbols import params as P
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn

dev = 'cuda:0'
cfg = lsfa_test_config(key_frame_interval=10)
arg, aux = P.init_params(cfg, seed=0)
cur = resnet_v1_101_flownet_rfcn(cfg).get_cur_test_symbol(cfg).bind(arg, aux, dev)
H, W = 600, 1000
data = torch.rand(1, 3, H, W, device=dev) * 255
im_info = torch.tensor([[H, W, 1.0]], device=dev)
feat = torch.randn(1, 1024, 38, 63, device=dev)
mv = torch.randn(1, 2, 38, 63, device=dev) * 0.5
res = torch.randn(1, 3, 38, 63, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
R, ncls = cfg.TEST.RPN_POST_NMS_TOP_N, cfg.dataset.NUM_CLASSES
bufs = (torch.zeros((ncls, R, 5), dtype=torch.float64, device=dev), torch.zeros(ncls, dtype=torch.int32, device=dev),
        torch.full((ncls, R), -1, dtype=torch.int32, device=dev))      # what core/graphs.py hands lsfa_det_postprocess: static buffers
if os.environ.get('LSFA_TUNED', '1') == '1':
    tuning.enable()
with torch.no_grad():
    for _ in range(n):
        out = cur.forward(data=data, im_info=im_info, feat_key=feat, motion_vector=mv, res_diff=res)
        hip.det_postprocess(out['rois_output'], out['bbox_pred_reshape_output'][0], out['cls_prob_reshape_output'][0], H, W, 1.0,
                            nms_thresh=cfg.TEST.NMS, max_per_image=cfg.TEST.max_per_image, class_agnostic=cfg.CLASS_AGNOSTIC, out=bufs)
torch.cuda.synchronize()
