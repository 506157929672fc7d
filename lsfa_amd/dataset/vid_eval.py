"""ImageNet-VID mAP@0.5 on detection rows.

Restates lib/dataset/imagenet_vid_eval.py: per-ground-truth IoU threshold
min(0.5, wh / ((w+10)(h+10))) (:34-37), per image greedy matching in confidence order with
`ov >= thr and ov > ovmax` (:165-193), AP as the area under the monotone precision envelope
(:45-67), mean over classes 1..C-1 (:205-218).  Input is in memory instead of the text file the
reference writes and re-parses; the writer below reproduces that file's format
(lib/dataset/imagenet_vid.py:266-268: '%d %d %.4f %.2f %.2f %.2f %.2f').
"""
import numpy as np


def gt_threshold(bbox, default_iou_thr=0.5, pixel_tolerance=10):
    w = bbox[2] - bbox[0] + 1
    h = bbox[3] - bbox[1] + 1
    return min((w * h) / ((w + pixel_tolerance) * (h + pixel_tolerance)), default_iou_thr)


def vid_ap(rec, prec):
    mrec = np.concatenate(([0.], rec, [1.]))
    mpre = np.concatenate(([0.], prec, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def format_rows(rows):
    """The reference's result-file lines; parsing them back is what vid_eval sees (values rounded
    to 4 / 2 decimals)."""
    return ['{:d} {:d} {:.4f} {:.2f} {:.2f} {:.2f} {:.2f}'.format(int(r[0]), int(r[1]), r[2], r[3], r[4], r[5], r[6])
            for r in rows]


def vid_eval(rows, gt, num_classes, through_text=True):
    """rows: (n,7) [frame_id, cls, score, x1,y1,x2,y2]; gt: list of dicts
    {'img_id': int, 'bbox': (k,4), 'label': (k,)}.  Returns ap[1:] like the reference."""
    if through_text and len(rows):
        rows = np.array([[float(z) for z in line.split(' ')] for line in format_rows(rows)])
    npos = np.zeros(num_classes)
    for rec in gt:
        for x in rec['label']:
            npos[int(x)] += 1
    gt_img_ids = [rec['img_id'] for rec in gt]
    if len(rows) == 0:
        return np.zeros(num_classes - 1)
    img_ids = rows[:, 0].astype(np.int64)
    order = np.argsort(img_ids, kind='stable')
    rows, img_ids = rows[order], img_ids[order]
    num_imgs = max(max(gt_img_ids), int(img_ids.max())) + 1
    cell = [None] * num_imgs
    for iid in np.unique(img_ids):
        sel = rows[img_ids == iid]
        cell[iid] = sel[np.argsort(-sel[:, 2], kind='stable')]
    tp_l, fp_l, lab_l, conf_l = [], [], [], []
    for rec in gt:
        det = cell[rec['img_id']]
        if det is None:
            continue
        gt_labels, gt_bboxes = np.asarray(rec['label']), np.asarray(rec['bbox'], dtype=np.float64).reshape(-1, 4)
        gt_thr = np.array([gt_threshold(b) for b in gt_bboxes])
        gt_detected = np.zeros(len(gt_labels))
        tp, fp = np.zeros(len(det)), np.zeros(len(det))
        for j in range(len(det)):
            bb, label = det[j, 3:7], int(det[j, 1])
            ovmax, kmax = -1, -1
            for k in range(len(gt_labels)):
                if label != gt_labels[k] or gt_detected[k] > 0:
                    continue
                bbgt = gt_bboxes[k]
                iw = min(bb[2], bbgt[2]) - max(bb[0], bbgt[0]) + 1
                ih = min(bb[3], bbgt[3]) - max(bb[1], bbgt[1]) + 1
                if iw > 0 and ih > 0:
                    ua = (bb[2] - bb[0] + 1.) * (bb[3] - bb[1] + 1.) + (bbgt[2] - bbgt[0] + 1.) * (bbgt[3] - bbgt[1] + 1.) - iw * ih
                    ov = iw * ih / ua
                    if ov >= gt_thr[k] and ov > ovmax:
                        ovmax, kmax = ov, k
            if kmax >= 0:
                tp[j] = 1
                gt_detected[kmax] = 1
            else:
                fp[j] = 1
        tp_l.append(tp); fp_l.append(fp); lab_l.append(det[:, 1].astype(np.int64)); conf_l.append(det[:, 2])
    if not tp_l:
        return np.zeros(num_classes - 1)
    tp_all, fp_all = np.concatenate(tp_l), np.concatenate(fp_l)
    labels, confs = np.concatenate(lab_l), np.concatenate(conf_l)
    order = np.argsort(-confs, kind='stable')
    tp_all, fp_all, labels = tp_all[order], fp_all[order], labels[order]
    ap = np.zeros(num_classes)
    for c in range(1, num_classes):
        fp = np.cumsum(fp_all[labels == c])
        tp = np.cumsum(tp_all[labels == c])
        rec = tp / float(npos[c]) if npos[c] > 0 else np.zeros_like(tp)
        prec = tp / np.maximum(tp + fp, np.finfo(np.float64).eps)
        ap[c] = vid_ap(rec, prec)
    return ap[1:]
