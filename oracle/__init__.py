"""CPU oracle for LSFA's per-frame hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  Nothing under ``lsfa_amd/`` does (a test enforces
it).  The arithmetic lives in ``lsfa_oracle.c`` (plain C, gcc); this module is
the numpy <-> C glue plus the pieces of the reference that are numpy in the
reference too (``np_ref.py``) and the torch-CPU statement of the conv graph
(``graph_ref.py``).

Pinning status (see lsfa_oracle.c header and DESIGN.md §Oracle):
  pinned by tests/golden/ref_golden.npz (generated from the reference's own numpy
  helpers by tests/golden/make_golden.py): anchors, IoU, NMS survivor lists, box
  decode/clip.  PARITY UNPINNED: warp, PSROI pooling, the decode/sort stages of
  Proposal, DCN, every conv — the reference holds no test or golden vector for
  them and their implementation is CUDA / un-vendored MXNet.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblsfa_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "lsfa_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liblsfa_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_dev_iou.restype = ctypes.c_float
        _lib.orc_nms_sorted.restype = ctypes.c_int
        _lib.orc_nms_f64.restype = ctypes.c_int
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


_ci = ctypes.c_int
_cf = ctypes.c_float
_cd = ctypes.c_double


def generate_anchors(feature_stride=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    r, s = _f32(ratios), _f32(scales)
    out = np.empty((len(r) * len(s), 4), np.float32)
    lib().orc_generate_anchors(_ci(feature_stride), _p(r), _ci(len(r)), _p(s), _ci(len(s)), _p(out))
    return out


def dev_iou(a, b):
    a, b = _f32(a), _f32(b)
    return float(lib().orc_dev_iou(_p(a), _p(b)))


def nms_sorted(boxes, thresh):
    """Bitmask-NMS semantics (nms_kernel.cu) on score-sorted float32 boxes (n, >=4)."""
    boxes = _f32(boxes)
    n, d = boxes.shape
    keep = np.empty(max(n, 1), np.int32)
    k = lib().orc_nms_sorted(_p(boxes), _ci(n), _ci(d), _cf(thresh), _p(keep))
    return keep[:k].copy()


def nms_mask(boxes, thresh):
    boxes = _f32(boxes)
    n, d = boxes.shape
    mask = np.zeros((n, (n + 63) // 64), np.uint64)
    lib().orc_nms_mask(_p(boxes), _ci(n), _ci(d), _cf(thresh), _p(mask))
    return mask


def gpu_nms(dets, thresh):
    """The recovered gpu_nms.pyx wrapper (lib/nms/gpu_nms.cu:1488-1490, 1727-1730,
    1772-1774): sort by score descending, run _nms on the sorted float32 boxes, map
    the survivors back through `order`.  Ties: ascending index (see np_ref.nms)."""
    dets = _f32(dets)
    order = np.argsort(-dets[:, 4], kind="stable")
    keep = nms_sorted(dets[order], thresh)
    return [int(i) for i in order[keep]]


def proposal_decode(cls_prob, bbox_pred, im_info, feature_stride=16, scales=(8, 16, 32),
                    ratios=(0.5, 1, 2), rpn_min_size=0):
    cls_prob, bbox_pred, im_info = _f32(cls_prob), _f32(bbox_pred), _f32(im_info)
    B, A2, H, W = cls_prob.shape
    A = A2 // 2
    s, r = _f32(scales), _f32(ratios)
    out = np.empty((B, H * W * A, 5), np.float32)
    lib().orc_proposal_decode(_p(cls_prob), _p(bbox_pred), _p(im_info), _ci(B), _ci(A), _ci(H), _ci(W),
                              _ci(feature_stride), _p(s), _ci(len(s)), _p(r), _ci(len(r)),
                              _ci(rpn_min_size), _p(out))
    return out


def proposal(cls_prob, bbox_pred, im_info, feature_stride=16, scales=(8, 16, 32), ratios=(0.5, 1, 2),
             rpn_pre_nms_top_n=6000, rpn_post_nms_top_n=300, threshold=0.7, rpn_min_size=0,
             return_debug=False):
    """MultiProposal (C++ semantics).  Returns rois (B*post_n,5), scores (B*post_n,1)."""
    cls_prob, bbox_pred, im_info = _f32(cls_prob), _f32(bbox_pred), _f32(im_info)
    B, A2, H, W = cls_prob.shape
    A = A2 // 2
    count = A * H * W
    pre_n = rpn_pre_nms_top_n if rpn_pre_nms_top_n > 0 else count
    pre_n = min(pre_n, count)
    post_n = min(rpn_post_nms_top_n, pre_n)
    s, r = _f32(scales), _f32(ratios)
    rois = np.empty((B * post_n, 5), np.float32)
    scores = np.empty((B * post_n, 1), np.float32)
    order = np.empty((B, pre_n), np.int32)
    keep = np.full((B, pre_n), -1, np.int32)
    nkeep = np.empty(B, np.int32)
    lib().orc_proposal(_p(cls_prob), _p(bbox_pred), _p(im_info), _ci(B), _ci(A), _ci(H), _ci(W),
                       _ci(feature_stride), _p(s), _ci(len(s)), _p(r), _ci(len(r)),
                       _ci(rpn_pre_nms_top_n), _ci(rpn_post_nms_top_n), _cf(threshold), _ci(rpn_min_size),
                       _p(rois), _p(scores), _p(order), _p(keep), _p(nkeep))
    if return_debug:
        return rois, scores, order, keep, nkeep
    return rois, scores


def psroi_pool(data, rois, spatial_scale, output_dim, pooled_size, group_size, with_mapping=False):
    data, rois = _f32(data), _f32(rois)
    N, C, H, W = data.shape
    R = rois.shape[0]
    out = np.empty((R, output_dim, pooled_size, pooled_size), np.float32)
    mc = np.empty_like(out) if with_mapping else None
    lib().orc_psroi_pool(_p(data), _p(rois), _ci(N), _ci(C), _ci(H), _ci(W), _ci(R), _cf(spatial_scale),
                         _ci(output_dim), _ci(pooled_size), _ci(group_size), _p(out), _p(mc))
    return (out, mc) if with_mapping else out


def global_avg(pooled):
    pooled = _f32(pooled)
    R, D, P, _ = pooled.shape
    out = np.empty((R, D), np.float32)
    lib().orc_global_avg(_p(pooled), _ci(R), _ci(D), _ci(P), _p(out))
    return out


def softmax_rows(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().orc_softmax_rows(_p(x), _ci(x.shape[0]), _ci(x.shape[1]), _p(y))
    return y


def rfcn_head(cls_map, box_map, rois, spatial_scale=0.0625, pooled_size=7, group_size=7):
    """psroipooled_* -> ave_* -> cls_prob  (resnet_v1_101_flownet_rfcn.py:520-540)."""
    ncls = cls_map.shape[1] // (group_size * group_size)
    nbox = box_map.shape[1] // (group_size * group_size)
    cls_score = global_avg(psroi_pool(cls_map, rois, spatial_scale, ncls, pooled_size, group_size))
    bbox_pred = global_avg(psroi_pool(box_map, rois, spatial_scale, nbox, pooled_size, group_size))
    return softmax_rows(cls_score), cls_score, bbox_pred


def warp_bilinear(feat, flow, mul=None, add=None, res=None, res_w=None, res_b=None):
    feat, flow = _f32(feat), _f32(flow)
    N, _, H, W = flow.shape
    feat_n, C = feat.shape[0], feat.shape[1]
    mul = _f32(mul) if mul is not None else None
    add = _f32(add) if add is not None else None
    res_c = 0
    if res is not None:
        res, res_w, res_b = _f32(res), _f32(res_w).reshape(C, -1), _f32(res_b)
        res_c = res.shape[1]
    out = np.empty((N, C, H, W), np.float32)
    lib().orc_warp_bilinear(_p(feat), _ci(feat_n), _p(flow), _ci(N), _ci(C), _ci(H), _ci(W), _p(mul), _p(add),
                            _p(res), _ci(res_c), _p(res_w), _p(res_b), _p(out))
    return out


def aggregate_softmax2(a, b, logits):
    """a, b (N,C,H,W), logits (2N,1,H,W): map n combines with logits rows n and N+n (resnet_v1_101_flownet_rfcn.py:95-108
    is the N = 1 case; for N > 1 every map is that case on its own pair of rows)."""
    a, b, logits = _f32(a), _f32(b), _f32(logits)
    N, C, H, W = a.shape
    out = np.empty_like(a)
    lg = logits.reshape(2, N, H, W)
    for n in range(N):
        pair = np.ascontiguousarray(lg[:, n])
        o = np.empty((1, C, H, W), np.float32)
        lib().orc_aggregate_softmax2(_p(np.ascontiguousarray(a[n])), _p(np.ascontiguousarray(b[n])), _p(pair), _ci(C), _ci(H),
                                     _ci(W), _p(o))
        out[n] = o[0]
    return out


def aggregate_cosine(a, b, emb_warp, emb_cur):
    a, b, emb_warp, emb_cur = _f32(a), _f32(b), _f32(emb_warp), _f32(emb_cur)
    _, C, H, W = a.shape
    E = emb_warp.shape[1]
    out = np.empty_like(a)
    lib().orc_aggregate_cosine(_p(a), _p(b), _p(emb_warp), _p(emb_cur), _ci(C), _ci(E), _ci(H), _ci(W), _p(out))
    return out


def bbox_pred_clip(rois, deltas, im_h, im_w, scale):
    rois, deltas = _f32(rois), _f32(deltas)
    R = rois.shape[0]
    nreg = deltas.shape[1] // 4
    out = np.empty((R, 4 * nreg), np.float64)
    lib().orc_bbox_pred_clip(_p(rois), _p(deltas), _ci(R), _ci(nreg), _cd(im_h), _cd(im_w), _cd(scale), _p(out))
    return out


def nms_f64(dets, thresh):
    dets = np.ascontiguousarray(dets, dtype=np.float64)
    n = dets.shape[0]
    keep = np.empty(max(n, 1), np.int32)
    k = lib().orc_nms_f64(_p(dets), _ci(n), _cd(thresh), _p(keep))
    return keep[:k].copy()


def det_postprocess(rois, deltas, probs, im_h, im_w, scale, score_thresh=1e-4, nms_thresh=0.3,
                    max_per_image=300, class_agnostic=True):
    rois, deltas, probs = _f32(rois), _f32(deltas), _f32(probs)
    R, ncls = probs.shape
    nreg = deltas.shape[1] // 4
    dets = np.zeros((ncls, R, 5), np.float64)
    counts = np.zeros(ncls, np.int32)
    keep_idx = np.full((ncls, R), -1, np.int32)
    lib().orc_det_postprocess(_p(rois), _p(deltas), _p(probs), _ci(R), _ci(ncls), _ci(nreg), _ci(int(class_agnostic)),
                              _cd(im_h), _cd(im_w), _cd(scale), _cd(score_thresh), _cd(nms_thresh),
                              _ci(max_per_image), _p(dets), _p(counts), _p(keep_idx))
    return dets, counts, keep_idx


def deform_im2col(data, offset, kh, kw, pad, stride, dilate, deform_groups):
    data, offset = _f32(data), _f32(offset)
    N, C, H, W = data.shape
    Ho, Wo = offset.shape[2], offset.shape[3]
    col = np.empty((N, C * kh * kw, Ho * Wo), np.float32)
    lib().orc_deform_im2col(_p(data), _p(offset), _ci(N), _ci(C), _ci(H), _ci(W), _ci(kh), _ci(kw), _ci(pad),
                            _ci(stride), _ci(dilate), _ci(deform_groups), _ci(Ho), _ci(Wo), _p(col))
    return col


def scale_shift_leaky(x, scale, shift, slope):
    x, scale, shift = _f32(x), _f32(scale), _f32(shift)
    N, C = x.shape[:2]
    HW = int(np.prod(x.shape[2:]))
    y = np.empty_like(x)
    lib().orc_scale_shift_leaky(_p(x), _p(scale), _p(shift), _ci(N), _ci(C), _ci(HW), _cf(slope), _p(y))
    return y


def scale_shift_relu(x, scale, shift, relu=True):
    x, scale, shift = _f32(x), _f32(scale), _f32(shift)
    N, C = x.shape[:2]
    HW = int(np.prod(x.shape[2:]))
    y = np.empty_like(x)
    lib().orc_scale_shift_relu(_p(x), _p(scale), _p(shift), _ci(N), _ci(C), _ci(HW), _ci(int(relu)), _p(y))
    return y


# ---- compressed-domain motion vectors (coviar_data_loader.c:71-177) -----------------------------
def coviar_identity(width, height):
    accu = np.empty((height, width, 2), np.int32)
    lib().orc_coviar_identity(_p(accu), _ci(width), _ci(height))
    return accu


def coviar_accumulate(mvs, accu_old):
    """One P-frame's macroblock motion vectors (n,7) int32 applied to the accumulated source map (H,W,2)."""
    mvs = np.ascontiguousarray(mvs, dtype=np.int32).reshape(-1, 7)
    accu_old = np.ascontiguousarray(accu_old, dtype=np.int32)
    h, w = accu_old.shape[:2]
    accu_new = accu_old.copy()
    lib().orc_coviar_accumulate(_p(mvs), _ci(mvs.shape[0]), _p(accu_old), _p(accu_new), _ci(w), _ci(h))
    return accu_new


def coviar_mv(accu):
    accu = np.ascontiguousarray(accu, dtype=np.int32)
    h, w = accu.shape[:2]
    mv = np.empty((h, w, 2), np.int32)
    lib().orc_coviar_mv(_p(accu), _ci(w), _ci(h), _p(mv))
    return mv


def coviar_residual(bgr_cur, bgr_ref, accu):
    bgr_cur, bgr_ref = np.ascontiguousarray(bgr_cur, np.uint8), np.ascontiguousarray(bgr_ref, np.uint8)
    accu = np.ascontiguousarray(accu, dtype=np.int32)
    h, w = accu.shape[:2]
    res = np.empty((h, w, 3), np.int32)
    lib().orc_coviar_residual(_p(bgr_cur), _p(bgr_ref), _p(accu), _ci(w), _ci(h), _p(res))
    return res
