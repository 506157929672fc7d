"""ImageNet-VID mAP@0.5 over in-memory detection rows.

Same measure as lib/dataset/imagenet_vid_eval.py — per-ground-truth IoU threshold
min(0.5, wh / ((w+10)(h+10))) (:34-37), greedy one-to-one assignment per image in confidence
order (:165-193), AP = area under the monotone precision envelope (:45-67), classes 1..C-1
(:205-218) — organised differently: matching never crosses classes, so the work is split into
independent (image, class) cells, each solved on a dense IoU matrix; the precision envelope is a
reversed running maximum.  Input is the gathered row tensor (no result file is written and
re-parsed; `format_rows` produces the reference's text lines, lib/dataset/imagenet_vid.py:266-268,
and `vid_eval(through_text=True)` rounds the values the way that file would).
tests/test_parallel_cpu.py checks it against the loop-form restatement oracle/np_ref.vid_eval_ref.
"""
import numpy as np

ROW_FORMAT = '{:d} {:d} {:.4f} {:.2f} {:.2f} {:.2f} {:.2f}'


def format_rows(rows):
    return [ROW_FORMAT.format(int(r[0]), int(r[1]), r[2], r[3], r[4], r[5], r[6]) for r in rows]


def gt_threshold(bbox, default_iou_thr=0.5, pixel_tolerance=10):
    """IoU a detection must reach to claim this ground-truth box: relaxed for small objects."""
    bw, bh = bbox[2] - bbox[0] + 1, bbox[3] - bbox[1] + 1
    return min(default_iou_thr, (bw * bh) / ((bw + pixel_tolerance) * (bh + pixel_tolerance)))


def overlap_matrix(dets, gts):
    """(n,4) x (k,4) -> (n,k) IoU with the +1 pixel convention; -1 where the boxes do not overlap."""
    lo = np.maximum(dets[:, None, :2], gts[None, :, :2])
    hi = np.minimum(dets[:, None, 2:], gts[None, :, 2:])
    span = hi - lo + 1.0
    touching = (span > 0).all(-1)
    inter = span[..., 0] * span[..., 1]
    area_d = (dets[:, 2] - dets[:, 0] + 1.0) * (dets[:, 3] - dets[:, 1] + 1.0)
    area_g = (gts[:, 2] - gts[:, 0] + 1.0) * (gts[:, 3] - gts[:, 1] + 1.0)
    iou = inter / (area_d[:, None] + area_g[None, :] - inter)
    return np.where(touching, iou, -1.0)


def assign(det_boxes, gt_boxes):
    """One (image, class) cell: detections in descending confidence.  -> bool (n,), True = matched a
    still-free ground truth whose threshold it reaches (the best such one, first on ties)."""
    hit = np.zeros(len(det_boxes), dtype=bool)
    if len(gt_boxes) == 0 or len(det_boxes) == 0:
        return hit
    need = np.array([gt_threshold(g) for g in gt_boxes])
    score = overlap_matrix(det_boxes, gt_boxes)
    score[score < need[None, :]] = -1.0
    free = np.ones(len(gt_boxes), dtype=bool)
    for i in range(len(det_boxes)):
        cand = np.where(free, score[i], -1.0)
        g = int(np.argmax(cand))
        if cand[g] > -1.0:
            hit[i] = True
            free[g] = False
    return hit


def average_precision(recall, precision):
    r = np.concatenate(([0.0], recall, [1.0]))
    p = np.concatenate(([0.0], precision, [0.0]))
    p = np.maximum.accumulate(p[::-1])[::-1]
    step = np.flatnonzero(r[1:] != r[:-1])
    return np.sum((r[step + 1] - r[step]) * p[step + 1])


def vid_eval(rows, gt, num_classes, through_text=True):
    """rows (n,7) = [frame_id, cls, score, x1, y1, x2, y2]; gt = [{'img_id', 'bbox' (k,4), 'label' (k,)}].
    -> AP of classes 1..num_classes-1.  Detections on frames without a gt record are ignored, like
    the reference's loop over the annotation list does."""
    rows = np.asarray(rows, dtype=np.float64).reshape(-1, 7)
    if through_text and len(rows):
        rows = np.array([[float(v) for v in line.split(' ')] for line in format_rows(rows)])
    positives = np.zeros(num_classes)
    for rec in gt:
        np.add.at(positives, np.asarray(rec['label'], dtype=np.int64), 1)
    if len(rows) == 0:
        return np.zeros(num_classes - 1)
    frame = rows[:, 0].astype(np.int64)
    by_frame = {}
    for f in np.unique(frame):
        block = rows[frame == f]                                  # file order
        by_frame[int(f)] = block[np.argsort(-block[:, 2], kind='stable')]
    # one entry per evaluated detection: (class, confidence, matched), in annotation-list order, then confidence order
    cls_l, conf_l, hit_l = [], [], []
    for rec in gt:
        det = by_frame.get(int(rec['img_id']))
        if det is None:
            continue
        labels = np.asarray(rec['label'], dtype=np.int64)
        boxes = np.asarray(rec['bbox'], dtype=np.float64).reshape(-1, 4)
        det_cls = det[:, 1].astype(np.int64)
        hit = np.zeros(len(det), dtype=bool)
        for c in np.unique(det_cls):
            sel = det_cls == c
            hit[sel] = assign(det[sel, 3:7], boxes[labels == c])
        cls_l.append(det_cls); conf_l.append(det[:, 2]); hit_l.append(hit)
    ap = np.zeros(num_classes)
    if not cls_l:
        return ap[1:]
    cls_all, conf_all, hit_all = np.concatenate(cls_l), np.concatenate(conf_l), np.concatenate(hit_l)
    rank = np.argsort(-conf_all, kind='stable')
    cls_all, hit_all = cls_all[rank], hit_all[rank]
    for c in range(1, num_classes):
        mine = hit_all[cls_all == c]
        tp, fp = np.cumsum(mine), np.cumsum(~mine)
        recall = tp / positives[c] if positives[c] > 0 else np.zeros(len(tp))
        precision = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
        ap[c] = average_precision(recall, precision)
    return ap[1:]
