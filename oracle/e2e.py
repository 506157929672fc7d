"""End-to-end parity criterion anchored on a float64 graph — TEST INFRASTRUCTURE ONLY (tests/, bench.py's parity leg).

north_star asks for "bit-exact ROI indices / NMS survivors, boxes and scores within 1e-4" between the GPU path and the
reference's fp32 CPU path.  Both are fp32 evaluations of ~100 layers with different summation orders, so neither is "the" answer:
two proposals whose RPN scores agree to the last bits can come out in either order, and a decoded corner of a 1000-px box carries
a few ulps (6.1e-5 px each) of accumulated round-off on either side.  Instead of widening constants this module measures BOTH
fp32 evaluations against the float64 statement of the same graph (oracle/graph_ref.py, dtype=float64) and asks:

  (a) is the GPU path's distance to float64 at most RATIO x the fp32 oracle's own distance (plus one fp32 ulp of the quantity),
      for the RPN scores and deltas, the ROI coordinates, the decoded boxes and the class probabilities;
  (b) where the two fp32 sides order two proposals differently, do the float64 scores of that pair differ by less than the fp32
      error bar of the score map (2 x the larger side's maximum score error)?  A swap that is not such a tie FAILS;
  (c) where a proposal survives Proposal's NMS on one side only, is its float64 IoU with a kept box within IOU_BAR of the
      threshold (a tie of the suppression test), or is it the lowest-ranked row that such a tie pushed over the post-NMS cap?
  (d) with ties matched by ANCHOR INDEX (multi_proposal.cu:57-67 enumeration, carried out of the Proposal stage), are the
      detection survivors (class, anchor) identical, apart from ROIs whose corner sits on a rounding boundary of PSROI pooling's
      round() (psroi_pooling.cu:56-59) within the two sides' coordinate error.
Nothing here is a tolerance on the GPU-vs-oracle distance itself; that distance is reported, and bounded by (a) through the
triangle inequality.
"""
import numpy as np

import oracle

RATIO = 1.5                     # GPU error <= RATIO x oracle-fp32 error (+ one ulp of the quantity: both sides are rounded to fp32)
ULP = {'rpn_score': 2.0 ** -24, 'rpn_delta': 2.0 ** -23, 'roi_px': 2.0 ** -14, 'box_px': 2.0 ** -14, 'cls_prob': 2.0 ** -24}
IOU_BAR = 4e-6                  # |IoU64 - thresh| below which Proposal's suppression test is a tie between fp32 evaluations
                                # (an IoU of two boxes whose corners carry ~1e-4 px of round-off on 100-1000 px extents)


def _np(a):
    return np.asarray(a)


def _half_away(v):
    return np.sign(v) * np.floor(np.abs(v) + 0.5)


def fg_scores(prob, A):
    """rpn_cls_prob (1, 2A, H, W) -> foreground score per anchor index ((h * W) + w) * A + a."""
    return _np(prob)[0, A:].transpose(1, 2, 0).reshape(-1)


def anchor_rows(anchors):
    """anchor index -> first output row holding it (the cyclic pad repeats rows)"""
    first = {}
    for i, a in enumerate(anchors):
        first.setdefault(int(a), i)
    return first


def image_of_batch(taps, out, b, B):
    """Image b of a frame pass that carried B images on the batch axis (the non-key frames of a segment, a group of key fronts), as the
    single-image taps / outputs the checks take: every tensor whose leading axis is B (or B x rows: MultiProposal's ROI layout, whose
    batch column is set to 0 like a single image's) is cut down to image b."""
    def cut(name, t):
        if not hasattr(t, 'shape') or t.dim() == 0:
            return t
        if name.endswith('_reshape_output') and t.dim() == 3 and t.shape[0] == 1 and t.shape[1] % B == 0:
            R = t.shape[1] // B              # (BATCH_IMAGES = 1, B * R, C): the batch symbol's shape (resnet_v1_101_flownet_rfcn.py:745-747)
            return t[:, b * R:(b + 1) * R]
        if name == 'nq_logits' and t.shape[0] == 2 * B:
            return t[[b, B + b]]             # rows b and B + b weigh image b's two maps (aggregate_softmax2's layout)
        if t.shape[0] == B:
            return t[b:b + 1]
        if name == 'rois_output' and t.shape[0] % B == 0:
            R = t.shape[0] // B
            r = t[b * R:(b + 1) * R].clone()
            r[:, 0] = 0
            return r
        return t
    return {k: cut(k, v) for k, v in taps.items()}, {k: cut(k, v) for k, v in out.items()}


def gpu_side(cfg, taps, out, im_info):
    """The GPU frame as a `side`: its own head maps + outputs, and the anchor index of every ROI row - taken from the oracle's
    Proposal run on the GPU's own maps, whose ROIs must equal the GPU's bit for bit (asserted here)."""
    f = lambda t: t.detach().float().cpu().numpy() if hasattr(t, 'detach') else np.asarray(t)
    prob, delta = f(taps['rpn_cls_prob']), f(taps['rpn_bbox_pred'])
    rois, _, order, keep, nkeep = oracle.proposal(prob, delta, im_info, cfg.network.RPN_FEAT_STRIDE, cfg.network.ANCHOR_SCALES,
                                                  cfg.network.ANCHOR_RATIOS, cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N,
                                                  cfg.TEST.RPN_NMS_THRESH, cfg.TEST.RPN_MIN_SIZE, return_debug=True)
    g_rois = f(out['rois_output'])
    if not np.array_equal(rois, g_rois):
        raise AssertionError("Proposal on the GPU differs from the oracle's on the GPU's own RPN maps: anchor identity undefined")
    from oracle.graph_ref import roi_anchor_index
    return dict(rpn_cls_prob=prob, rpn_bbox_pred=delta, rois_output=g_rois, roi_anchor=roi_anchor_index(order, keep, nkeep, rois.shape[0]),
                cls_prob_reshape_output=f(out['cls_prob_reshape_output']), bbox_pred_reshape_output=f(out['bbox_pred_reshape_output']))


def _decoded64(cfg, side64, im_info):
    """all anchors' decoded boxes of the float64 graph (fp32 decode of its rounded maps, like the operator)"""
    d = oracle.proposal_decode(side64['rpn_cls_prob'], side64['rpn_bbox_pred'], im_info, cfg.network.RPN_FEAT_STRIDE,
                               cfg.network.ANCHOR_SCALES, cfg.network.ANCHOR_RATIOS, cfg.TEST.RPN_MIN_SIZE)
    return np.asarray(d[0, :, :4], np.float64)


def _iou64(a, b):
    iw = np.minimum(a[2], b[:, 2]) - np.maximum(a[0], b[:, 0]) + 1.0
    ih = np.minimum(a[3], b[:, 3]) - np.maximum(a[1], b[:, 1]) + 1.0
    inter = np.maximum(iw, 0.0) * np.maximum(ih, 0.0)
    sa = (a[2] - a[0] + 1.0) * (a[3] - a[1] + 1.0)
    sb = (b[:, 2] - b[:, 0] + 1.0) * (b[:, 3] - b[:, 1] + 1.0)
    return inter / (sa + sb - inter)


def frame_gap(cfg, gpu, ref32, ref64, im_info, h, w):
    """gpu / ref32 / ref64: sides (dicts with rpn_cls_prob, rpn_bbox_pred, rois_output, roi_anchor, cls_prob_reshape_output,
    bbox_pred_reshape_output).  -> dict of measurements and `failures` (list of strings, empty when the criterion holds)."""
    A = cfg.network.NUM_ANCHORS
    fail = []
    rec = {}
    s_g, s_r, s_64 = (fg_scores(x['rpn_cls_prob'], A).astype(np.float64) for x in (gpu, ref32, ref64))
    err = {'rpn_score': (float(np.abs(s_g - s_64).max()), float(np.abs(s_r - s_64).max()))}
    d64 = _np(ref64['rpn_bbox_pred']).astype(np.float64)
    err['rpn_delta'] = (float(np.abs(_np(gpu['rpn_bbox_pred']) - d64).max()), float(np.abs(_np(ref32['rpn_bbox_pred']) - d64).max()))
    bar = 2.0 * max(err['rpn_score'])

    a_g, a_r, a_64 = (np.asarray(x['roi_anchor'], np.int64) for x in (gpu, ref32, ref64))
    set_g, set_r = set(a_g.tolist()), set(a_r.tolist())
    R = len(a_g)

    # (c) proposals kept on one side only: a tie of the suppression test, or the last row pushed over the cap by one
    only = sorted(set_g ^ set_r)
    explained_iou, pushed = [], []
    if only:
        boxes64 = _decoded64(cfg, ref64, im_info)
        kept_union = np.array(sorted(set_g | set_r), np.int64)
        last_rows = {int(a_g[len(set_g) - 1]) if len(set_g) else -1, int(a_r[len(set_r) - 1]) if len(set_r) else -1}
        for x in only:
            others = kept_union[kept_union != x]
            iou = _iou64(boxes64[x], boxes64[others])
            close = float(np.abs(iou - cfg.TEST.RPN_NMS_THRESH).min()) if len(others) else 1.0
            if close <= IOU_BAR:
                explained_iou.append((int(x), close))
            elif x in last_rows and len(explained_iou) > 0:
                pushed.append(int(x))
            else:
                fail.append("anchor %d survives Proposal on one side only and no float64 IoU is within %.1e of the threshold (closest %.3e)"
                            % (x, IOU_BAR, close))
    rec.update(roi_only_one_side=len(only), roi_iou_ties=len(explained_iou), roi_pushed_over_cap=len(pushed))

    # (b) the order of the common proposals: position by position among the anchors both sides kept
    common = set_g & set_r
    seq_g = [int(a) for a in a_g[:len(set_g)] if int(a) in common]
    seq_r = [int(a) for a in a_r[:len(set_r)] if int(a) in common]
    displaced, ties, worst_gap = 0, 0, 0.0
    for x, y in zip(seq_g, seq_r):
        if x != y:
            displaced += 1
            gap = abs(s_64[x] - s_64[y])
            worst_gap = max(worst_gap, gap)
            if gap <= bar:
                ties += 1
            else:
                fail.append("ROIs ordered differently (anchors %d / %d) but their float64 scores differ by %.3e > fp32 error bar %.3e" % (x, y, gap, bar))
    rec.update(roi_displaced=displaced, roi_displaced_ties=ties, roi_displaced_worst_f64_gap=worst_gap, score_error_bar=bar,
               roi_order_identical=bool(np.array_equal(a_g, a_r)))

    # (a) row quantities on the SAME sample for both sides: anchors all three graphs kept, compared by anchor index
    rows = [anchor_rows(a) for a in (a_g, a_r, a_64)]
    S = sorted(set(rows[0]) & set(rows[1]) & set(rows[2]))
    ig, ir, i64 = (np.array([r[a] for a in S], np.int64) for r in rows)
    roi_g, roi_r, roi_64 = (_np(x['rois_output'])[i][:, 1:].astype(np.float64) for x, i in ((gpu, ig), (ref32, ir), (ref64, i64)))
    err['roi_px'] = (float(np.abs(roi_g - roi_64).max()), float(np.abs(roi_r - roi_64).max())) if S else (0.0, 0.0)
    # PSROI pooling rounds the ROI corners: a corner within the sides' coordinate error of a .5 boundary pools different bins -
    # a discontinuity of the reference's own operator.  Those anchors are counted and left out of the box / probability distances.
    stable = (_half_away(roi_g) == _half_away(roi_64)).all(1) & (_half_away(roi_r) == _half_away(roi_64)).all(1)
    near = np.abs(np.abs(np.concatenate([roi_g, roi_r], 1)) % 1.0 - 0.5).min(1)
    for k in np.nonzero(~stable)[0]:
        if near[k] > 4.0 * max(err['roi_px']) + 1e-6:
            fail.append("anchor %d: ROI corners round differently although none is within the coordinate error of a .5 boundary" % S[k])
    rec['unstable_rois'] = int((~stable).sum())
    box = lambda x, i: oracle.bbox_pred_clip(_np(x['rois_output'])[i], _np(x['bbox_pred_reshape_output'])[0][i], h, w, 1.0)
    cls = lambda x, i: _np(x['cls_prob_reshape_output'])[0][i].astype(np.float64)
    if stable.any():
        b_g, b_r, b_64 = box(gpu, ig)[stable], box(ref32, ir)[stable], box(ref64, i64)[stable]
        c_g, c_r, c_64 = cls(gpu, ig)[stable], cls(ref32, ir)[stable], cls(ref64, i64)[stable]
        err['box_px'] = (float(np.abs(b_g - b_64).max()), float(np.abs(b_r - b_64).max()))
        err['cls_prob'] = (float(np.abs(c_g - c_64).max()), float(np.abs(c_r - c_64).max()))
        rec['max_abs_dbox'] = float(np.abs(b_g - b_r).max())            # GPU vs the fp32 oracle, reported (north_star's quantity)
        rec['max_abs_dscore'] = float(np.abs(c_g - c_r).max())
    for k, (eg, er) in err.items():
        # `ratio` is the plain quotient; `ratio_beyond_one_ulp` takes the criterion's one-ulp allowance off the GPU side first - for quantities
        # whose errors are one to three fp32 ulps (a class probability near 1: 6e-8 each) the plain quotient is a ratio of small integers
        rec['err_vs_f64_' + k] = dict(gpu=eg, oracle_fp32=er, ratio=(eg / er if er > 0 else None),
                                      ratio_beyond_one_ulp=(max(eg - ULP[k], 0.0) / er if er > 0 else None), ulp=ULP[k])
        if eg > RATIO * er + ULP[k]:
            fail.append("%s: GPU is %.3e from float64, the fp32 oracle %.3e (allowed %.1f x + %.1e)" % (k, eg, er, RATIO, ULP[k]))
    rec['rois_compared'] = len(S)

    # (d) detection survivors as (class, anchor) pairs
    kw = dict(nms_thresh=cfg.TEST.NMS, max_per_image=cfg.TEST.max_per_image, class_agnostic=cfg.CLASS_AGNOSTIC)

    def survivors(x, anchors):
        _, c, k = oracle.det_postprocess(_np(x['rois_output']), _np(x['bbox_pred_reshape_output'])[0], _np(x['cls_prob_reshape_output'])[0], h, w, 1.0, **kw)
        return set((j, int(anchors[i])) for j in range(1, len(c)) for i in k[j, :c[j]])
    sv_g, sv_r = survivors(gpu, a_g), survivors(ref32, a_r)
    excused = set(int(S[k]) for k in np.nonzero(~stable)[0]) | set(only)
    diff = sv_g ^ sv_r
    rec.update(survivors=len(sv_r), survivor_mismatch=len(diff))
    # an unstable / one-sided ROI may change the survivors of the classes it competes in; anything else must agree
    if len(diff) > 8 * len(excused):
        fail.append("%d detection survivors differ with %d excusable ROIs" % (len(diff), len(excused)))
    if not excused and diff:
        fail.append("detection survivors differ: %s" % sorted(diff)[:6])
    if not sv_r:
        fail.append("no detections survive on the oracle side: the comparison is vacuous")
    rec['failures'] = fail
    return rec
