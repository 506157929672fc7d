"""Shared pieces of the graph-level parity tests (GPU side vs the oracle).

`check_key_frame` / `check_cur_frame` pin every hand-written stage of ONE frame to the oracle,
bit for bit, by feeding the oracle the GPU's own inputs to that stage (teacher forcing): a
1-ulp difference in a library convolution can then neither hide nor fake a kernel bug, and a
frame that was served a stale key feature, a stale small-net feature or another frame's
motion vectors fails `array_equal` instead of slipping under a tolerance.
"""
import json
import os

import numpy as np
import torch

import oracle


def np_(t):
    return t.detach().float().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def check_heads(cfg, taps, out, im_info):
    """Proposal + PSROI/avg/softmax of one frame, bit-exact on the GPU's own head maps."""
    rois, _ = oracle.proposal(np_(taps['rpn_cls_prob']), np_(taps['rpn_bbox_pred']), im_info, cfg.network.RPN_FEAT_STRIDE,
                              cfg.network.ANCHOR_SCALES, cfg.network.ANCHOR_RATIOS, cfg.TEST.RPN_PRE_NMS_TOP_N,
                              cfg.TEST.RPN_POST_NMS_TOP_N, cfg.TEST.RPN_NMS_THRESH, cfg.TEST.RPN_MIN_SIZE)
    np.testing.assert_array_equal(np_(out['rois_output']), rois)
    cls_prob, _, bbox_pred = oracle.rfcn_head(np_(taps['cls_map']), np_(taps['box_map']), rois)
    np.testing.assert_array_equal(np_(out['cls_prob_reshape_output'])[0], cls_prob)
    np.testing.assert_array_equal(np_(out['bbox_pred_reshape_output'])[0], bbox_pred)


def check_dets(cfg, out, dets, counts, h, w, scale=1.0, thresh=1e-4):
    """lsfa_det_postprocess of one frame vs the oracle's pred_eval post-processing on the same network outputs."""
    wd, wc, _ = oracle.det_postprocess(np_(out['rois_output']), np_(out['bbox_pred_reshape_output'])[0],
                                       np_(out['cls_prob_reshape_output'])[0], h, w, scale, score_thresh=thresh,
                                       nms_thresh=cfg.TEST.NMS, max_per_image=cfg.TEST.max_per_image,
                                       class_agnostic=cfg.CLASS_AGNOSTIC)
    counts = np.asarray(counts)
    np.testing.assert_array_equal(counts, wc)
    dets = np.asarray(dets)
    for j in range(1, len(wc)):
        np.testing.assert_array_equal(dets[j, :wc[j]], wd[j, :wc[j]])


def check_key_frame(cfg, taps, out, feat_key_old, im_info):
    """Second-and-later key frame: flow warp x scale map of the PREVIOUS key feature, Nq aggregation, heads."""
    warp_want = oracle.warp_bilinear(np_(feat_key_old), np_(taps['flow']), mul=np_(taps['scale_map']))
    np.testing.assert_array_equal(np_(taps['warp']), warp_want)
    agg_want = oracle.aggregate_softmax2(warp_want, np_(taps['backbone_feat']), np_(taps['nq_logits']))
    np.testing.assert_array_equal(np_(out['choose_feat_output']), agg_want)
    check_heads(cfg, taps, out, im_info)


def check_cur_frame(cfg, arg, taps, out, feat_key, mv, res, im_info):
    """Non-key frame: MV warp of ITS key frame's feature + rnet_conv0(res_diff) + small-net feature, heads."""
    want = oracle.warp_bilinear(np_(feat_key), np_(mv), add=np_(taps['small_feat']), res=np_(res),
                                res_w=arg['rnet_conv0_weight'], res_b=arg['rnet_conv0_bias'])
    np.testing.assert_array_equal(np_(out['conv_feat']), want)
    check_heads(cfg, taps, out, im_info)


def assert_dets_equal(d0, c0, d1, c1, msg=""):
    """Detections of two runs agree exactly: same per-class counts, same rows (rows past a class's count are
    whatever an earlier frame left in that buffer and are not part of the result)."""
    d0, c0, d1, c1 = np_d(d0), np_d(c0), np_d(d1), np_d(c1)
    np.testing.assert_array_equal(c0, c1, err_msg=msg + " (per-class counts)")
    for j in range(1, len(c0)):
        np.testing.assert_array_equal(d0[j, :c0[j]], d1[j, :c0[j]], err_msg=msg + " (class %d)" % j)


def np_d(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def clone_dict(d, keys=None):
    return {k: v.clone() for k, v in d.items() if isinstance(v, torch.Tensor) and (keys is None or k in keys)}


def record(name, payload):
    """Measured tolerances etc. for the round's notes: gpurun_out/parity_<name>.json (scratch, merged back by gpurun)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_%s.json" % name), "w") as f:
            json.dump(payload, f, indent=1, sort_keys=True)
    except OSError:
        pass
