"""The reference's frame loop on the CPU oracle — TEST INFRASTRUCTURE ONLY.

pred_eval (dff_rfcn/core/tester.py:237-281) + TestLoader's flag schedule
(dff_rfcn/core/loader.py:87-141) over oracle/graph_ref.py.  `frames(f)`, `mv(f, key_f)`, `res(f)`
are callables returning numpy arrays; returns detection rows [frame_id, cls, score, x1,y1,x2,y2]."""
import numpy as np

import oracle
from oracle import graph_ref, np_ref


def run_clip(cfg, arg, aux, num_frames, frames, mv, res, im_info, frame_id0=0, thresh=1e-4):
    flags = np_ref.key_frame_flags([num_frames], cfg.TEST.KEY_FRAME_INTERVAL)
    H, W = int(im_info[0, 0]), int(im_info[0, 1])
    rows = []
    feat = data_key = None
    key_f = 0
    for f, flag in enumerate(flags):
        data = frames(f)
        if flag != 2:
            feat_old = np.zeros((1, 1024, 1, 1), np.float32) if flag == 0 else feat
            data_key_old = data if (flag == 0 or data_key is None) else data_key
            out = graph_ref.key_forward(cfg, arg, aux, data, data_key_old, feat_old, im_info)
            feat, data_key, key_f = out['choose_feat_output'], data, f
        else:
            out = graph_ref.cur_forward(cfg, arg, aux, data, feat, mv(f, key_f), res(f), im_info)
        dets, counts, _ = oracle.det_postprocess(out['rois_output'], out['bbox_pred_reshape_output'][0],
                                                 out['cls_prob_reshape_output'][0], H, W, float(im_info[0, 2]),
                                                 score_thresh=thresh, nms_thresh=cfg.TEST.NMS,
                                                 max_per_image=cfg.TEST.max_per_image, class_agnostic=cfg.CLASS_AGNOSTIC)
        for j in range(1, dets.shape[0]):
            for k in range(counts[j]):
                rows.append([frame_id0 + f, j, dets[j, k, 4]] + list(dets[j, k, :4]))
    return np.asarray(rows, dtype=np.float64).reshape(-1, 7)
