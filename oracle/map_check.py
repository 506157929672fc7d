"""mAP agreement between a GPU run and the CPU oracle — TEST INFRASTRUCTURE ONLY (tests/, bench.py's parity leg).

north_star: "mAP on the demo clip within 0.1 of the reference".  No trained weights or annotations exist here, so the
metric is the one tests/test_graph_gpu.py::test_clip_end_to_end_map_vs_oracle introduced: the ORACLE's most confident detections of every
frame are taken as ground truth, and VID mAP@0.5 (lsfa_amd/dataset/vid_eval.py, checked against the loop-form restatement of
lib/dataset/imagenet_vid_eval.py:70-218) is computed for the oracle's own rows (the ceiling: < 1 because only the top detections are
ground truth) and for the GPU's rows; the agreement number is their difference.

The frame schedule is bench.py's: frame 0 is the clip's first frame (flag 0), then intervals of K frames starting at frames 1, K + 1, ...
whose first frame is a key frame (flag 1: warped + aggregated with the previous key frame's feature) and whose others are non-key (flag 2)."""
import numpy as np

import oracle
from oracle import graph_ref


def _rows(out, cfg, H, W, scale, frame, thresh=1e-4):
    dets, counts, _ = oracle.det_postprocess(out['rois_output'], out['bbox_pred_reshape_output'][0], out['cls_prob_reshape_output'][0], H, W, scale,
                                             score_thresh=thresh, nms_thresh=cfg.TEST.NMS, max_per_image=cfg.TEST.max_per_image,
                                             class_agnostic=cfg.CLASS_AGNOSTIC)
    rows = []
    for j in range(1, dets.shape[0]):
        for k in range(counts[j]):
            rows.append([frame, j, dets[j, k, 4]] + list(dets[j, k, :4]))
    return rows


def oracle_rows(cfg, arg, aux, frames, mv, res, im_info, K, intervals, known=None):
    """Detection rows [frame, cls, score, x1, y1, x2, y2] of frames 1 .. intervals*K on the fp32 oracle graph.
    frames(f) -> (1, 3, H, W); mv(f) / res(f) -> the non-key frame's tensors; known: {frame: graph_ref output} already computed (reused)."""
    H, W, scale = int(im_info[0, 0]), int(im_info[0, 1]), float(im_info[0, 2])
    known = dict(known or {})
    if 0 not in known:
        f0 = frames(0)
        known[0] = graph_ref.key_forward(cfg, arg, aux, f0, f0, np.zeros((1, 1024, 1, 1), np.float32), im_info)
    rows, prev_key = [], 0
    for s in range(intervals):
        kf = 1 + s * K
        if kf not in known:
            known[kf] = graph_ref.key_forward(cfg, arg, aux, frames(kf), frames(prev_key), known[prev_key]['choose_feat_output'], im_info)
        rows += _rows(known[kf], cfg, H, W, scale, kf)
        for i in range(1, K):
            f = kf + i
            if f not in known:
                known[f] = graph_ref.cur_forward(cfg, arg, aux, frames(f), known[kf]['choose_feat_output'], mv(f), res(f), im_info)
            rows += _rows(known[f], cfg, H, W, scale, f)
            known.pop(f)                 # a non-key frame's output is not needed again
        prev_key = kf
    return np.asarray(rows, dtype=np.float64).reshape(-1, 7)


def map_vs_oracle(rows_gpu, rows_ref, frame_ids, num_classes=31, top=5):
    """-> {"map_gpu", "map_oracle", "delta", ...}: VID mAP@0.5 of both row sets against the oracle's `top` most confident detections per frame"""
    from lsfa_amd.dataset import vid_eval as ve
    gt = []
    for f in frame_ids:
        r = rows_ref[rows_ref[:, 0] == f]
        r = r[np.argsort(-r[:, 2], kind='stable')][:top]
        gt.append({'img_id': f, 'bbox': r[:, 3:7], 'label': r[:, 1].astype(int)})
    ap_ref = ve.vid_eval(rows_ref, gt, num_classes)
    ap_gpu = ve.vid_eval(rows_gpu, gt, num_classes)
    present = np.unique(np.concatenate([g['label'] for g in gt])) - 1
    m_ref, m_gpu = float(ap_ref[present].mean()), float(ap_gpu[present].mean())
    return {"map_gpu": round(m_gpu, 4), "map_oracle": round(m_ref, 4), "delta": round(m_gpu - m_ref, 4), "frames": len(list(frame_ids)),
            "classes_present": int(len(present)), "rows_gpu": int(len(rows_gpu)), "rows_oracle": int(len(rows_ref)),
            "ground_truth": "the fp32 oracle's %d most confident detections per frame (oracle/map_check.py)" % top}
