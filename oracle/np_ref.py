"""numpy restatement of the reference's numpy helpers — TEST INFRASTRUCTURE ONLY.

These follow the reference line by line where the reference itself is numpy, so
the golden vectors captured from the reference (tests/golden/ref_golden.npz) pin
them directly.  Paths are relative to the hustvl/LSFA tree.
"""
import numpy as np


# ---- lib/rpn/generate_anchor.py:21-85 ------------------------------------------------
def _whctrs(anchor):
    w = anchor[2] - anchor[0] + 1
    h = anchor[3] - anchor[1] + 1
    return w, h, anchor[0] + 0.5 * (w - 1), anchor[1] + 0.5 * (h - 1)


def _mkanchors(ws, hs, x_ctr, y_ctr):
    ws, hs = ws[:, np.newaxis], hs[:, np.newaxis]
    return np.hstack((x_ctr - 0.5 * (ws - 1), y_ctr - 0.5 * (hs - 1),
                      x_ctr + 0.5 * (ws - 1), y_ctr + 0.5 * (hs - 1)))


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=2 ** np.arange(3, 6)):
    ratios, scales = np.asarray(ratios, dtype=np.float64), np.asarray(scales)
    base_anchor = np.array([1, 1, base_size, base_size]) - 1
    w, h, x_ctr, y_ctr = _whctrs(base_anchor)
    size_ratios = (w * h) / ratios
    ws = np.round(np.sqrt(size_ratios))
    hs = np.round(ws * ratios)
    ratio_anchors = _mkanchors(ws, hs, x_ctr, y_ctr)
    out = []
    for i in range(ratio_anchors.shape[0]):
        w, h, x_ctr, y_ctr = _whctrs(ratio_anchors[i, :])
        out.append(_mkanchors(w * scales, h * scales, x_ctr, y_ctr))
    return np.vstack(out)


# ---- lib/nms/nms.py:37-74 -------------------------------------------------------------
def nms(dets, thresh):
    """Greedy NMS, keeps `ovr <= thresh`.  The reference orders with
    `scores.argsort()[::-1]`, whose tie order is whatever numpy's unstable sort
    yields; this restatement fixes ties as (score desc, index asc)."""
    if dets.shape[0] == 0:
        return []
    x1, y1, x2, y2, scores = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = np.argsort(-scores, kind="stable")
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        xx1 = np.maximum(x1[i], x1[order[1:]])
        yy1 = np.maximum(y1[i], y1[order[1:]])
        xx2 = np.minimum(x2[i], x2[order[1:]])
        yy2 = np.minimum(y2[i], y2[order[1:]])
        w = np.maximum(0.0, xx2 - xx1 + 1)
        h = np.maximum(0.0, yy2 - yy1 + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[order[1:]] - inter)
        inds = np.where(ovr <= thresh)[0]
        order = order[inds + 1]
    return keep


# ---- lib/bbox/bbox_transform.py:45-60, :103-140 -----------------------------------------
def clip_boxes(boxes, im_shape):
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], im_shape[1] - 1), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], im_shape[0] - 1), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], im_shape[1] - 1), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], im_shape[0] - 1), 0)
    return boxes


def bbox_pred(boxes, box_deltas):
    if boxes.shape[0] == 0:
        return np.zeros((0, box_deltas.shape[1]))
    boxes = boxes.astype(np.float64, copy=False)
    widths = boxes[:, 2] - boxes[:, 0] + 1.0
    heights = boxes[:, 3] - boxes[:, 1] + 1.0
    ctr_x = boxes[:, 0] + 0.5 * (widths - 1.0)
    ctr_y = boxes[:, 1] + 0.5 * (heights - 1.0)
    dx, dy, dw, dh = box_deltas[:, 0::4], box_deltas[:, 1::4], box_deltas[:, 2::4], box_deltas[:, 3::4]
    pred_ctr_x = dx * widths[:, np.newaxis] + ctr_x[:, np.newaxis]
    pred_ctr_y = dy * heights[:, np.newaxis] + ctr_y[:, np.newaxis]
    pred_w = np.exp(dw) * widths[:, np.newaxis]
    pred_h = np.exp(dh) * heights[:, np.newaxis]
    pred_boxes = np.zeros(box_deltas.shape)
    pred_boxes[:, 0::4] = pred_ctr_x - 0.5 * (pred_w - 1.0)
    pred_boxes[:, 1::4] = pred_ctr_y - 0.5 * (pred_h - 1.0)
    pred_boxes[:, 2::4] = pred_ctr_x + 0.5 * (pred_w - 1.0)
    pred_boxes[:, 3::4] = pred_ctr_y + 0.5 * (pred_h - 1.0)
    return pred_boxes


def bbox_overlaps(boxes, query_boxes):
    """bbox_overlaps_py, lib/bbox/bbox_transform.py:22-42 (vectorised)."""
    n_, k_ = boxes.shape[0], query_boxes.shape[0]
    overlaps = np.zeros((n_, k_), dtype=np.float64)
    for k in range(k_):
        qa = (query_boxes[k, 2] - query_boxes[k, 0] + 1) * (query_boxes[k, 3] - query_boxes[k, 1] + 1)
        for n in range(n_):
            iw = min(boxes[n, 2], query_boxes[k, 2]) - max(boxes[n, 0], query_boxes[k, 0]) + 1
            if iw > 0:
                ih = min(boxes[n, 3], query_boxes[k, 3]) - max(boxes[n, 1], query_boxes[k, 1]) + 1
                if ih > 0:
                    ba = (boxes[n, 2] - boxes[n, 0] + 1) * (boxes[n, 3] - boxes[n, 1] + 1)
                    overlaps[n, k] = iw * ih / float(ba + qa - iw * ih)
    return overlaps


# ---- dff_rfcn/core/tester.py:143-152, :265-281 ------------------------------------------
def im_detect_post(rois, scores, bbox_deltas, im_shape, scale):
    """rois (R,5) float32, bbox_deltas (R, 4*nreg) float32 -> pred_boxes float64."""
    pred_boxes = bbox_pred(rois[:, 1:], bbox_deltas)
    pred_boxes = clip_boxes(pred_boxes, im_shape[-2:])
    return scores, pred_boxes / scale


def pred_eval_post(scores, boxes, num_classes, thresh=1e-4, nms_thresh=0.3, max_per_image=300,
                   class_agnostic=True):
    """Returns all_boxes[j] = (n_j, 5) float64 for j in 0..num_classes-1 (class 0 empty)."""
    all_boxes = [np.zeros((0, 5))] * num_classes
    for j in range(1, num_classes):
        indexes = np.where(scores[:, j] > thresh)[0]
        cls_scores = scores[indexes, j, np.newaxis]
        cls_boxes = boxes[indexes, 4:8] if class_agnostic else boxes[indexes, j * 4:(j + 1) * 4]
        cls_dets = np.hstack((cls_boxes, cls_scores))
        keep = nms(cls_dets, nms_thresh)
        all_boxes[j] = cls_dets[keep, :]
    if max_per_image > 0:
        image_scores = np.hstack([all_boxes[j][:, -1] for j in range(1, num_classes)])
        if len(image_scores) > max_per_image:
            image_thresh = np.sort(image_scores)[-max_per_image]
            for j in range(1, num_classes):
                keep = np.where(all_boxes[j][:, -1] >= image_thresh)[0]
                all_boxes[j] = all_boxes[j][keep, :]
    return all_boxes


# ---- dff_rfcn/core/loader.py:87-141 ------------------------------------------------------
def key_frame_flags(seg_lens, key_frame_interval):
    """Flag sequence TestLoader emits: 0 first frame of a video, 1 key frame
    (every KEY_FRAME_INTERVAL and the last frame of each video), 2 non-key."""
    flags = []
    for L in seg_lens:
        key_frameid = 0
        for f in range(L):
            if key_frameid == f:
                flags.append(0 if key_frameid == 0 else 1)
            elif f + 1 == L:
                flags.append(1)
            else:
                flags.append(2)
            nf = f + 1
            if nf == L:
                pass
            elif nf - key_frameid == key_frame_interval:
                key_frameid = nf
    return flags


# ---- dff_rfcn/function/test_rcnn.py:69-75 ------------------------------------------------
def shard_videos(seg_lens, gpu_num):
    shards = [[] for _ in range(gpu_num)]
    acc = np.zeros(gpu_num, dtype=np.int64)
    for vid, L in enumerate(seg_lens):
        g = int(np.argmin(acc))
        shards[g].append(vid)
        acc[g] += L
    return shards


# ---- lib/utils/image.py:202-308 with cv2.resize(INTER_LINEAR) restated -------------------
def cv2_resize_linear(src, fx, fy):
    """cv2.resize(src, None, None, fx, fy, INTER_LINEAR) for float images (OpenCV 3.2 resize.cpp:
    dsize = cvRound(size*f); with fx / fy given the scale is 1/f (`scale_x = 1. / inv_scale_x`), NOT
    src/dst; source coordinate (d + 0.5)*scale - 0.5, clamped so that the two taps stay inside;
    horizontal pass then vertical pass) [un-vendored, parity unpinned].
    NOT restated: at fx == fy == 0.5 exactly, OpenCV 3.2's resize() switches INTER_LINEAR to INTER_AREA (resizeAreaFast,
    (a + b + c + d) * 0.25f).  The two taps below weigh the same 2 x 2 block by 0.25 each: equal bit for bit on
    integer-valued sources (frames, motion vectors, residuals), up to an ulp apart on fractional float32 maps."""
    # resize.cpp's linear_tab: CV_32F runs HResizeLinear<float, float, float> / VResizeLinear<float, float, float>, CV_64F
    # HResizeLinear<double, double, float> / VResizeLinear<double, double, float>: a float64 image (the zero-padded maps of
    # transform_mv_res, image.py:210-222) is interpolated in double with the SAME float coefficients and comes back float64
    src = np.asarray(src)
    wt = np.float64 if src.dtype == np.float64 else np.float32
    src = src.astype(wt)
    sh, sw = src.shape[:2]
    dh, dw = int(np.rint(sh * fy)), int(np.rint(sw * fx))

    def taps(dn, sn, factor):
        scale = 1.0 / float(factor)
        # resize.cpp: `fx = (float)((dx+0.5)*scale_x - 0.5); sx = cvFloor(fx); fx -= sx;` — the position is rounded to
        # float BEFORE the floor and the subtraction
        f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s0 = np.floor(f).astype(np.int64)
        a = (f - s0.astype(np.float32)).astype(np.float32)
        lo = s0 < 0
        s0[lo], a[lo] = 0, 0.0
        hi = s0 >= sn - 1
        s0[hi], a[hi] = sn - 1, 0.0
        return s0, np.minimum(s0 + 1, sn - 1), a

    x0, x1, ax = taps(dw, sw, fx)
    y0, y1, ay = taps(dh, sh, fy)
    src3 = src.reshape(sh, sw, -1)
    bx, by = (1 - ax).astype(wt), (1 - ay).astype(wt)           # `cbuf[0] = 1.f - fx` in float, then used in the work type
    ax, ay = ax.astype(wt), ay.astype(wt)
    hor = src3[:, x0] * bx[None, :, None] + src3[:, x1] * ax[None, :, None]
    out = hor[y0] * by[:, None, None] + hor[y1] * ay[:, None, None]
    return out.astype(wt).reshape((dh, dw) + src.shape[2:])


def cv2_resize_linear_u8(src, fx, fy):
    """cv2.resize(src, None, None, fx, fy, INTER_LINEAR) for a UINT8 image: OpenCV 3.2's fixed-point path (imgwarp.cpp) - what the
    reference runs on the LAST frame of a video, which lib/utils/image.py:45 reads with cv2.imread [un-vendored, parity unpinned:
    restated from the published source, no OpenCV in this image to pin it].  Taps and positions as in the float path; the two
    coefficients of an axis become shorts `saturate_cast<short>(c * 2048)` (cvRound: ties to even) of c = 1.f - f and f (float);
    horizontal pass in int32, `D = S[x0] * a0 + S[x1] * a1`; vertical pass
    `dst = uchar((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2)` (VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>>:
    the form its SSE2 kernel computes with 16-bit high multiplies).  NOT restated: the switch to INTER_AREA at fx == fy == 0.5."""
    src = np.asarray(src)
    assert src.dtype == np.uint8
    sh, sw = src.shape[:2]
    dh, dw = int(np.rint(sh * fy)), int(np.rint(sw * fx))

    def taps(dn, sn, factor):
        scale = 1.0 / float(factor)
        f = ((np.arange(dn, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s0 = np.floor(f).astype(np.int64)
        a = (f - s0.astype(np.float32)).astype(np.float32)
        lo = s0 < 0
        s0[lo], a[lo] = 0, 0.0
        hi = s0 >= sn - 1
        s0[hi], a[hi] = sn - 1, 0.0
        c0 = (np.float32(1.0) - a).astype(np.float32)
        i0 = np.rint(c0 * np.float32(2048.0)).astype(np.int32)       # saturate_cast<short>(float) = cvRound (nearest, ties to even); <= 2048
        i1 = np.rint(a * np.float32(2048.0)).astype(np.int32)
        return s0, np.minimum(s0 + 1, sn - 1), i0, i1

    x0, x1, a0, a1 = taps(dw, sw, fx)
    y0, y1, b0, b1 = taps(dh, sh, fy)
    s3 = src.reshape(sh, sw, -1).astype(np.int32)
    hor = s3[:, x0] * a0[None, :, None] + s3[:, x1] * a1[None, :, None]
    out = (((b0[:, None, None] * (hor[y0] >> 4)) >> 16) + ((b1[:, None, None] * (hor[y1] >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8).reshape((dh, dw) + src.shape[2:])


def transform_mv_res(motion_vector, res_diff, im_scale, pixel_means, pixel_scale, rcnn_stride=16):
    motion_vector = cv2_resize_linear(motion_vector.astype(np.float32), im_scale, im_scale)
    res_diff = cv2_resize_linear(res_diff.astype(np.float32), im_scale, im_scale)
    im_h, im_w, _ = res_diff.shape
    p_im_h = int(np.ceil(im_h / float(rcnn_stride)) * rcnn_stride)
    p_im_w = int(np.ceil(im_w / float(rcnn_stride)) * rcnn_stride)
    padded_motion_vector = np.zeros((p_im_h, p_im_w, 2))
    padded_res_diff = np.zeros((p_im_h, p_im_w, 3))
    padded_motion_vector[:im_h, :im_w] = motion_vector
    padded_res_diff[:im_h, :im_w] = res_diff
    for i in range(3):
        padded_res_diff[:, :, i] = (padded_res_diff[:, :, 2 - i] - pixel_means[2 - i]) * pixel_scale
    rcnn_scale = 1.0 / rcnn_stride
    resize_motion_vector = cv2_resize_linear(padded_motion_vector, rcnn_scale, rcnn_scale)      # float64 in, float64 out
    resize_res_diff = cv2_resize_linear(padded_res_diff, rcnn_scale, rcnn_scale)
    resize_motion_vector *= im_scale * rcnn_scale
    th, tw, _ = resize_res_diff.shape
    return (resize_motion_vector.transpose((2, 0, 1)).reshape(1, 2, th, tw),
            resize_res_diff.transpose((2, 0, 1)).reshape(1, 3, th, tw))


def resize_scale(im_shape, target_size, max_size):
    im_size_min, im_size_max = np.min(im_shape[0:2]), np.max(im_shape[0:2])
    im_scale = float(target_size) / float(im_size_min)
    if np.round(im_scale * im_size_max) > max_size:
        im_scale = float(max_size) / float(im_size_max)
    return im_scale


def transform(im, pixel_means, pixel_scale):
    # config.network.PIXEL_MEANS is a list of Python numbers once update_network_config has run (dff_rfcn/config/config.py:172-182; the yaml's
    # too).  A float32 image - the decoder's frame after get_image's .astype(np.float32) and cv2.resize (lib/utils/image.py:52, 283) - minus a
    # Python float is a FLOAT32 subtraction (mean rounded to float32 first) under the reference's numpy and under today's alike; a uint8 image
    # (cv2.imread: the last frame of a video, :45) minus a Python float is float64.  Coerced to Python floats so that the restatement does not
    # depend on whether a test passes a list, a tuple or an array (an np.float64 ELEMENT would make numpy >= 2 subtract in float64).
    pixel_means = [float(m) for m in pixel_means]
    im_tensor = np.zeros((1, 3, im.shape[0], im.shape[1]))
    for i in range(3):
        im_tensor[0, i, :, :] = im[:, :, 2 - i] - pixel_means[2 - i]
    return im_tensor * pixel_scale


# ---- VID mAP, loop by loop as lib/dataset/imagenet_vid_eval.py states it (checker for
# lsfa_amd/dataset/vid_eval.py, which is vectorised): threshold :34-37, AP :45-67, per-image greedy
# matching :165-193, per-class accumulation :205-218 --------------------------------------------
def vid_gt_threshold(bbox, default_iou_thr=0.5, pixel_tolerance=10):
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


def vid_format_rows(rows):
    """The reference's result-file lines; parsing them back is what vid_eval sees (values rounded
    to 4 / 2 decimals)."""
    return ['{:d} {:d} {:.4f} {:.2f} {:.2f} {:.2f} {:.2f}'.format(int(r[0]), int(r[1]), r[2], r[3], r[4], r[5], r[6])
            for r in rows]


def vid_eval_ref(rows, gt, num_classes, through_text=True):
    """rows: (n,7) [frame_id, cls, score, x1,y1,x2,y2]; gt: list of dicts
    {'img_id': int, 'bbox': (k,4), 'label': (k,)}.  Returns ap[1:] like the reference."""
    if through_text and len(rows):
        rows = np.array([[float(z) for z in line.split(' ')] for line in vid_format_rows(rows)])
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
        gt_thr = np.array([vid_gt_threshold(b) for b in gt_bboxes])
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
