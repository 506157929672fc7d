"""ctypes binding of liblsfa_hip.so (the C ABI in include/lsfa_hip.h) on torch tensors.

PyTorch is plumbing here: it owns device memory and streams; every op below hands
raw device pointers and torch's CURRENT stream to the C ABI.  There is no CPU or
eager fallback — if the library is missing or a call fails, LsfaError is raised
(the counterpart of MXNetError in the reference).
"""
import ctypes
import math
import os

import torch  # must be imported before the .so so that both share one libamdhip64

_HERE = os.path.dirname(os.path.abspath(__file__))
# LSFA_HIP_LIBRARY: another build of the same library (tools/lab/drop_product_check.sh runs the parity suite against a deliberately
# degraded one to show that the suite notices); there is no fallback either way
LIB_PATH = os.environ.get("LSFA_HIP_LIBRARY") or os.path.join(_HERE, "liblsfa_hip.so")

# names of the LSFA_OP_* ids, in id order.  The table is the LIBRARY's (lsfa_op_name) — filled in by lib(); this copy only serves
# code that runs before the library is loaded and is checked against it there (an out-of-date copy made lsfa_prof_read write one
# element past the buffers sized by it: r3, after LSFA_OP_FLOWNET was added)
OP_NAMES = ["psroi_pool", "rfcn_head", "warp_bilinear", "aggregate", "proposal", "nms", "det_postprocess",
            "deform_im2col", "scale_shift_relu", "conv_nhwc", "stem", "flownet_small"]


class LsfaError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            # not a fallback: build the real library if the toolchain is here, else fail loudly
            try:
                from lsfa_amd import build as _build
                _build.build_hip()
            except Exception as e:
                raise LsfaError("liblsfa_hip.so is not built (%s) and building it failed (%r); run "
                                "`python -m lsfa_amd.build` — there is no fallback path" % (LIB_PATH, e))
        L = ctypes.CDLL(LIB_PATH)
        L.lsfa_last_error.restype = ctypes.c_char_p
        for name in ("lsfa_proposal_workspace_bytes", "lsfa_nms_workspace_bytes", "lsfa_det_workspace_bytes",
                     "lsfa_mv_workspace_bytes", "lsfa_conv_nhwc_workspace_bytes", "lsfa_conv_weight_bytes",
                     "lsfa_conv_workspace_bytes", "lsfa_deconv4x4s2_crop_workspace_bytes", "lsfa_stem_weight_bytes"):
            getattr(L, name).restype = ctypes.c_size_t
        L.lsfa_op_name.restype = ctypes.c_char_p
        L._nms.restype = None
        names, i = [], 0
        while True:
            nm = L.lsfa_op_name(ctypes.c_int(i))
            if not nm or nm == b"?" or i >= 64:          # "?" is what the library returns past the last id
                break
            names.append(nm.decode())
            i += 1
        if names != OP_NAMES:
            OP_NAMES[:] = names              # in place: importers of the list see the library's table
        wv = os.environ.get("LSFA_WARP_VARIANT")         # A/B runs of whole programs: 'gather' | 'staged' | 'auto'
        if wv:
            variants = {'auto': 0, 'gather': 1, 'staged': 2}
            if wv not in variants:
                raise LsfaError("LSFA_WARP_VARIANT=%r: expected one of %s" % (wv, sorted(variants)))
            L.lsfa_warp_set_variant(ctypes.c_int(variants[wv]))
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise LsfaError("%s failed (code %d): %s" % (what, rc, lib().lsfa_last_error().decode()))


_vp = ctypes.c_void_p
_ci = ctypes.c_int
_cf = ctypes.c_float
_cd = ctypes.c_double


def _ptr(t):
    return _vp(t.data_ptr()) if t is not None else None


def _stream():
    return _vp(torch.cuda.current_stream().cuda_stream)


def _on_tensor_device(fn):
    """Run an op with the device of its first tensor argument current, so that `_stream()` (the CURRENT
    stream of the CURRENT device) and the C side's per-device state belong to the tensors' device even
    when the caller never called torch.cuda.set_device (e.g. Predictor(..., context='cuda:1'))."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kw):
        dev = next((a.device for a in args if isinstance(a, torch.Tensor) and a.is_cuda), None)
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kw)
        with torch.cuda.device(dev):
            return fn(*args, **kw)
    return wrapped


def _f32c(t, name):
    if t is None:
        return None
    if t.dtype != torch.float32 or not t.is_cuda:
        raise LsfaError("%s must be a float32 CUDA tensor (got %s on %s)" % (name, t.dtype, t.device))
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------
@_on_tensor_device
def psroi_pool(data, rois, spatial_scale, output_dim, pooled_size, group_size, with_mapping=False):
    data, rois = _f32c(data, "data"), _f32c(rois, "rois")
    N, C, H, W = data.shape
    R = rois.shape[0]
    out = torch.empty((R, output_dim, pooled_size, pooled_size), device=data.device, dtype=torch.float32)
    mc = torch.empty_like(out) if with_mapping else None
    _check(lib().lsfa_psroi_pool_fwd(_ptr(data), _ptr(rois), _ci(N), _ci(C), _ci(H), _ci(W), _ci(R),
                                     _cf(spatial_scale), _ci(output_dim), _ci(pooled_size), _ci(group_size),
                                     _ptr(out), _ptr(mc), _stream()), "lsfa_psroi_pool_fwd")
    return (out, mc) if with_mapping else out


@_on_tensor_device
def rfcn_head(cls_map, box_map, rois, spatial_scale=0.0625, pooled_size=7, group_size=7, want_score=False,
              out=None):
    cls_map, box_map, rois = _f32c(cls_map, "cls_map"), _f32c(box_map, "box_map"), _f32c(rois, "rois")
    N, Cc, H, W = cls_map.shape
    gg = group_size * group_size
    ncls, nbox = Cc // gg, box_map.shape[1] // gg
    R = rois.shape[0]
    if out is None:
        cls_prob = torch.empty((R, ncls), device=rois.device, dtype=torch.float32)
        bbox_pred = torch.empty((R, nbox), device=rois.device, dtype=torch.float32)
    else:
        cls_prob, bbox_pred = out
    cls_score = torch.empty((R, ncls), device=rois.device, dtype=torch.float32) if want_score else None
    _check(lib().lsfa_rfcn_head_fwd(_ptr(cls_map), _ptr(box_map), _ptr(rois), _ci(N), _ci(H), _ci(W), _ci(R),
                                    _ci(ncls), _ci(nbox), _cf(spatial_scale), _ci(pooled_size), _ci(group_size),
                                    _ptr(cls_prob), _ptr(cls_score), _ptr(bbox_pred), _stream()), "lsfa_rfcn_head_fwd")
    return (cls_prob, cls_score, bbox_pred) if want_score else (cls_prob, bbox_pred)


@_on_tensor_device
def rfcn_head_ps(ps_map, rois, ncls, nbox, spatial_scale=0.0625, pooled_size=7, group_size=7, want_score=False):
    """ps_map (N, H, W, group^2, ncls+nbox) float32 — see lsfa_rfcn_head_ps_fwd."""
    ps_map, rois = _f32c(ps_map, "ps_map"), _f32c(rois, "rois")
    N, H, W = ps_map.shape[0], ps_map.shape[1], ps_map.shape[2]
    R = rois.shape[0]
    cls_prob = torch.empty((R, ncls), device=rois.device, dtype=torch.float32)
    bbox_pred = torch.empty((R, nbox), device=rois.device, dtype=torch.float32)
    cls_score = torch.empty((R, ncls), device=rois.device, dtype=torch.float32) if want_score else None
    _check(lib().lsfa_rfcn_head_ps_fwd(_ptr(ps_map), _ptr(rois), _ci(N), _ci(H), _ci(W), _ci(R), _ci(ncls), _ci(nbox),
                                       _cf(spatial_scale), _ci(pooled_size), _ci(group_size), _ptr(cls_prob),
                                       _ptr(cls_score), _ptr(bbox_pred), _stream()), "lsfa_rfcn_head_ps_fwd")
    return (cls_prob, cls_score, bbox_pred) if want_score else (cls_prob, bbox_pred)


@_on_tensor_device
def warp_bilinear(feat, flow, mul=None, add=None, res=None, res_w=None, res_b=None, out=None):
    feat, flow = _f32c(feat, "feat"), _f32c(flow, "flow")
    mul, add, res = _f32c(mul, "mul"), _f32c(add, "add"), _f32c(res, "res")
    N, _, H, W = flow.shape
    feat_n, C = feat.shape[0], feat.shape[1]
    res_c = 0
    if res is not None:
        res_w, res_b = _f32c(res_w, "res_w").reshape(C, -1), _f32c(res_b, "res_b")
        res_c = res.shape[1]
    if out is None:
        out = torch.empty((N, C, H, W), device=feat.device, dtype=torch.float32)
    _check(lib().lsfa_warp_bilinear(_ptr(feat), _ci(feat_n), _ptr(flow), _ci(N), _ci(C), _ci(H), _ci(W), _ptr(mul),
                                    _ptr(add), _ptr(res), _ci(res_c), _ptr(res_w), _ptr(res_b), _ptr(out), _stream()),
           "lsfa_warp_bilinear")
    return out


@_on_tensor_device
def warp_bilinear_cl(feat_cl, flow, add_cl=None, res=None, res_w=None, res_b=None, out=None, amax_out=None, amax_c0=0):
    """lsfa_warp_bilinear_cl: the non-key path's warp on channels-last maps.  feat_cl (feat_n, H, W, C), flow (N, 2, H, W), add_cl (N, H, W, C),
    res (N, res_c, H, W) -> (N, H, W, C); the bits of warp_bilinear on the transposed maps.  amax_out: a zeroed row of amax_slots() that
    receives max|out| over channels [amax_c0, C) (the scale of the convolution that reads those)."""
    feat_cl, flow = _f32c(feat_cl, "feat_cl"), _f32c(flow, "flow")
    add_cl, res = _f32c(add_cl, "add_cl"), _f32c(res, "res")
    N, _, H, W = flow.shape
    feat_n, C = feat_cl.shape[0], feat_cl.shape[3]
    if tuple(feat_cl.shape[1:3]) != (H, W) or (add_cl is not None and tuple(add_cl.shape) != (N, H, W, C)):
        raise LsfaError("warp_bilinear_cl: feat_cl %s / add_cl %s do not match a (%d, 2, %d, %d) flow" % (
            tuple(feat_cl.shape), None if add_cl is None else tuple(add_cl.shape), N, H, W))
    res_c = 0
    if res is not None:
        res_w, res_b = _f32c(res_w, "res_w").reshape(C, -1), _f32c(res_b, "res_b")
        res_c = res.shape[1]
    if out is None:
        out = torch.empty((N, H, W, C), device=feat_cl.device, dtype=torch.float32)
    _check(lib().lsfa_warp_bilinear_cl(_ptr(feat_cl), _ci(feat_n), _ptr(flow), _ci(N), _ci(C), _ci(H), _ci(W), _ptr(add_cl), _ptr(res), _ci(res_c),
                                       _ptr(res_w), _ptr(res_b), _ptr(out), _ptr(amax_out), _ci(amax_c0), _stream()), "lsfa_warp_bilinear_cl")
    return out


def warp_set_variant(variant):
    """'auto' | 'gather' | 'staged': lsfa_warp_set_variant (process-wide kernel choice of warp_bilinear; same results.  'staged'
    makes shapes the LDS-staged kernel does not take an error instead of falling back)."""
    _check(lib().lsfa_warp_set_variant(_ci({'auto': 0, 'gather': 1, 'staged': 2}[variant])), "lsfa_warp_set_variant")


@_on_tensor_device
def aggregate_softmax2(a, b, logits, out=None, logit_row_stride=None):
    """a, b (N, C, H, W); logits (2N, 1, H, W): rows [0, N) weight a, rows [N, 2N) weight b.
    logit_row_stride (r5): `logits` is any float32 tensor whose 2N rows of H*W logits start that many floats apart (the Nq net's
    last convolution written NCHW with 64 padded output channels: channel 0 of each image, stride 64*H*W) - no strided copy."""
    a, b = _f32c(a, "a"), _f32c(b, "b")
    N, C, H, W = a.shape
    if out is None:
        out = torch.empty_like(a)
    if logit_row_stride is not None:
        if (logits.dtype != torch.float32 or not logits.is_contiguous() or logits.device != a.device or logit_row_stride < H * W or
                logits.numel() < (2 * N - 1) * logit_row_stride + H * W):
            raise LsfaError("aggregate_softmax2: logits %s (%s, %s) must be contiguous float32 on %s with %d rows of %d logits >= %d floats apart, got stride %d" % (
                tuple(logits.shape), logits.dtype, logits.device, a.device, 2 * N, H * W, H * W, logit_row_stride))
        _check(lib().lsfa_aggregate_softmax2_rows(_ptr(a), _ptr(b), _ptr(logits), ctypes.c_long(logit_row_stride), _ci(N), _ci(C), _ci(H), _ci(W),
                                                  _ptr(out), _stream()), "lsfa_aggregate_softmax2_rows")
        return out
    logits = _f32c(logits, "logits")
    if logits.numel() != 2 * N * H * W:
        raise LsfaError("aggregate_softmax2: logits %s do not match a %s" % (tuple(logits.shape), tuple(a.shape)))
    if N == 1:
        _check(lib().lsfa_aggregate_softmax2(_ptr(a), _ptr(b), _ptr(logits), _ci(C), _ci(H), _ci(W), _ptr(out), _stream()),
               "lsfa_aggregate_softmax2")
    else:
        _check(lib().lsfa_aggregate_softmax2_batched(_ptr(a), _ptr(b), _ptr(logits), _ci(N), _ci(C), _ci(H), _ci(W),
                                                     _ptr(out), _stream()), "lsfa_aggregate_softmax2_batched")
    return out


@_on_tensor_device
def aggregate_cosine(a, b, emb_warp, emb_cur, out=None):
    a, b = _f32c(a, "a"), _f32c(b, "b")
    emb_warp, emb_cur = _f32c(emb_warp, "emb_warp"), _f32c(emb_cur, "emb_cur")
    _, C, H, W = a.shape
    if out is None:
        out = torch.empty_like(a)
    _check(lib().lsfa_aggregate_cosine(_ptr(a), _ptr(b), _ptr(emb_warp), _ptr(emb_cur), _ci(C), _ci(emb_warp.shape[1]),
                                       _ci(H), _ci(W), _ptr(out), _stream()), "lsfa_aggregate_cosine")
    return out


class ProposalOp(object):
    """MultiProposal / Proposal (multi_proposal-inl.h:124-159 params).  The workspace comes from torch's
    caching allocator per call, so calls issued on different streams (or captured into different
    graphs) never share scratch memory."""

    def __init__(self, feature_stride=16, scales=(8, 16, 32), ratios=(0.5, 1, 2), rpn_pre_nms_top_n=6000,
                 rpn_post_nms_top_n=300, threshold=0.7, rpn_min_size=16, output_score=False):
        self.feature_stride = int(feature_stride)
        self.scales = (ctypes.c_float * len(scales))(*[float(s) for s in scales])
        self.ratios = (ctypes.c_float * len(ratios))(*[float(r) for r in ratios])
        self.pre_n, self.post_n = int(rpn_pre_nms_top_n), int(rpn_post_nms_top_n)
        self.threshold, self.min_size = float(threshold), int(rpn_min_size)
        self.output_score = output_score

    def __call__(self, cls_prob, bbox_pred, im_info, out=None):
        if cls_prob.is_cuda and cls_prob.device.index != torch.cuda.current_device():
            with torch.cuda.device(cls_prob.device):
                return self(cls_prob, bbox_pred, im_info, out)
        cls_prob, bbox_pred, im_info = _f32c(cls_prob, "cls_prob"), _f32c(bbox_pred, "bbox_pred"), _f32c(im_info, "im_info")
        B, A2, H, W = cls_prob.shape
        A = A2 // 2
        if bbox_pred.shape != (B, 4 * A, H, W) or im_info.shape != (B, 3):
            raise LsfaError("Proposal: expected bbox_pred %s and im_info %s, got %s and %s (multi_proposal-inl.h:179-187)"
                            % ((B, 4 * A, H, W), (B, 3), tuple(bbox_pred.shape), tuple(im_info.shape)))
        need = lib().lsfa_proposal_workspace_bytes(_ci(B), _ci(A), _ci(H), _ci(W), _ci(self.pre_n))
        ws = torch.empty(need, dtype=torch.uint8, device=cls_prob.device)
        count = A * H * W
        pre_n = min(self.pre_n if self.pre_n > 0 else count, count)
        post_n = min(self.post_n, pre_n)
        if out is None:
            rois = torch.empty((B * post_n, 5), device=cls_prob.device, dtype=torch.float32)
            scores = torch.empty((B * post_n, 1), device=cls_prob.device, dtype=torch.float32)
        else:
            rois, scores = out
        _check(lib().lsfa_proposal(_ptr(cls_prob), _ptr(bbox_pred), _ptr(im_info), _ci(B), _ci(A), _ci(H), _ci(W),
                                   _ci(self.feature_stride), self.scales, _ci(len(self.scales)), self.ratios,
                                   _ci(len(self.ratios)), _ci(self.pre_n), _ci(self.post_n), _cf(self.threshold),
                                   _ci(self.min_size), _ptr(rois), _ptr(scores), _ptr(ws),
                                   ctypes.c_size_t(ws.numel()), _stream()), "lsfa_proposal")
        return (rois, scores) if self.output_score else rois


@_on_tensor_device
def proposal_set_plan(plan):
    """'auto' | 'single' | 'chip' | 'chip-box-sweep': lsfa_proposal_set_plan (process-wide launch plan; same results)."""
    _check(lib().lsfa_proposal_set_plan(_ci({'auto': 0, 'single': 1, 'chip': 2, 'chip-box-sweep': 3, 'lab-no-nms': 99}[plan])), "lsfa_proposal_set_plan")


# Ablation switches (tools/lab/tail_ablation.sh): LSFA_LAB_SKIP=tail,copy,nhwc drops launches from every frame so that
# bench.py's frames/s with and without them says what they cost the PIPELINE (not what they take alone):
#   tail  the Proposal's suppression stage (nms_mask + nms_sweep: `nms`), the PSROI head (`head`), the detection post-processing (det_class +
#         det_cap: `det`); each part also by its own name
#   copy  lsfa_copy_many's staging copies of frames / motion vectors / residuals into a graph's static buffers (after each buffer's first fill)
#   nhwc  lsfa_nchw_to_nhwc in front of the R-FCN / Nq convolutions (after each output's first fill)
# Results are garbage then; bench.py labels the line, reports no parity and does not read the status word.  Never set in a product run.
LAB_SKIP = set(v for v in os.environ.get('LSFA_LAB_SKIP', '').split(',') if v)
LAB_SKIP_TAIL = 'tail' in LAB_SKIP
LAB_SKIP_NMS = LAB_SKIP_TAIL or 'nms' in LAB_SKIP        # ... or one part of the tail: nms | head | det
LAB_SKIP_HEAD = LAB_SKIP_TAIL or 'head' in LAB_SKIP
LAB_SKIP_DET = LAB_SKIP_TAIL or 'det' in LAB_SKIP
_lab_filled = set()


def nms_sorted(boxes, thresh):
    """boxes (n, >=4) float32 CUDA, sorted by score descending -> (keep int32 (n,), num_keep int32 (1,)) on device."""
    boxes = _f32c(boxes, "boxes")
    n, d = boxes.shape
    keep = torch.empty(max(n, 1), dtype=torch.int32, device=boxes.device)
    num = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    need = lib().lsfa_nms_workspace_bytes(_ci(n))
    ws = torch.empty(need, dtype=torch.uint8, device=boxes.device)
    _check(lib().lsfa_nms_sorted(_ptr(boxes), _ci(n), _ci(d), _cf(thresh), _ptr(keep), _ptr(num), _ptr(ws),
                                 ctypes.c_size_t(need), _stream()), "lsfa_nms_sorted")
    return keep, num


@_on_tensor_device
def nms_sorted_f64(boxes, thresh):
    """boxes (n, >=4) float64 CUDA, sorted by score descending -> (keep, num_keep) like nms_sorted; numpy's
    float64 arithmetic and `ovr <= thresh` keep rule (lib/nms/nms.py:47-72)."""
    if boxes.dtype != torch.float64 or not boxes.is_cuda:
        raise LsfaError("boxes must be a float64 CUDA tensor (got %s on %s)" % (boxes.dtype, boxes.device))
    boxes = boxes.contiguous()
    n, d = boxes.shape
    keep = torch.empty(max(n, 1), dtype=torch.int32, device=boxes.device)
    num = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    need = lib().lsfa_nms_workspace_bytes(_ci(n))
    ws = torch.empty(need, dtype=torch.uint8, device=boxes.device)
    _check(lib().lsfa_nms_sorted_f64(_ptr(boxes), _ci(n), _ci(d), _cd(thresh), _ptr(keep), _ptr(num), _ptr(ws),
                                     ctypes.c_size_t(need), _stream()), "lsfa_nms_sorted_f64")
    return keep, num


def nms_host(boxes_np, thresh, device_id=0):
    """The reference's `_nms` C entry point (lib/nms/gpu_nms.hpp:14-15): host numpy in, host list out."""
    import numpy as np
    boxes_np = np.ascontiguousarray(boxes_np, dtype=np.float32)
    n, d = boxes_np.shape
    keep = np.zeros(max(n, 1), dtype=np.int32)
    num = ctypes.c_int(0)
    lib()._nms(keep.ctypes.data_as(_vp), ctypes.byref(num), boxes_np.ctypes.data_as(_vp), _ci(n), _ci(d),
               _cf(thresh), _ci(device_id))
    return keep[:num.value]


@_on_tensor_device
def bbox_pred_clip(rois, deltas, im_h, im_w, scale):
    rois, deltas = _f32c(rois, "rois"), _f32c(deltas, "deltas")
    R = rois.shape[0]
    nreg = deltas.shape[1] // 4
    out = torch.empty((R, 4 * nreg), dtype=torch.float64, device=rois.device)
    _check(lib().lsfa_bbox_pred_clip(_ptr(rois), _ptr(deltas), _ci(R), _ci(nreg), _cd(im_h), _cd(im_w), _cd(scale),
                                     _ptr(out), _stream()), "lsfa_bbox_pred_clip")
    return out


@_on_tensor_device
def det_postprocess(rois, deltas, probs, im_h, im_w, scale, score_thresh=1e-4, nms_thresh=0.3, max_per_image=300,
                    class_agnostic=True, out=None):
    rois, deltas, probs = _f32c(rois, "rois"), _f32c(deltas, "deltas"), _f32c(probs, "probs")
    R, ncls = probs.shape
    nreg = deltas.shape[1] // 4
    if out is None:
        dets = torch.zeros((ncls, R, 5), dtype=torch.float64, device=rois.device)
        counts = torch.zeros(ncls, dtype=torch.int32, device=rois.device)
        keep_idx = torch.full((ncls, R), -1, dtype=torch.int32, device=rois.device)
    else:
        dets, counts, keep_idx = out
    if LAB_SKIP_DET:
        return dets, counts, keep_idx
    _check(lib().lsfa_det_postprocess(_ptr(rois), _ptr(deltas), _ptr(probs), _ci(R), _ci(ncls), _ci(nreg),
                                      _ci(int(class_agnostic)), _cd(im_h), _cd(im_w), _cd(scale), _cd(score_thresh),
                                      _cd(nms_thresh), _ci(max_per_image), _ptr(dets), _ptr(counts), _ptr(keep_idx),
                                      None, ctypes.c_size_t(0), _stream()), "lsfa_det_postprocess")
    return dets, counts, keep_idx


@_on_tensor_device
def rpn_softmax_split(logits, A):
    """lsfa_rpn_softmax_split: logits (N, H, W, L >= 6A) channels-last [score 2A | delta 4A | padding] -> (rpn_cls_prob (N, 2A, H, W),
    rpn_bbox_pred (N, 4A, H, W)): the per-anchor two-way softmax and the split into MultiProposal's NCHW inputs."""
    logits = _f32c(logits, "logits")
    N, H, W, L = logits.shape
    cls_prob = torch.empty((N, 2 * A, H, W), device=logits.device, dtype=torch.float32)
    bbox = torch.empty((N, 4 * A, H, W), device=logits.device, dtype=torch.float32)
    _check(lib().lsfa_rpn_softmax_split(_ptr(logits), _ci(N), _ci(H), _ci(W), _ci(L), _ci(A), _ptr(cls_prob), _ptr(bbox), _stream()),
           "lsfa_rpn_softmax_split")
    return cls_prob, bbox


@_on_tensor_device
def det_postprocess_batch(rois, deltas, probs, B, im_h, im_w, scale, out, score_thresh=1e-4, nms_thresh=0.3, max_per_image=300,
                          class_agnostic=True):
    """lsfa_det_postprocess_batch: det_postprocess for the B images of a batch in one launch pair.  rois (B*R, 5), deltas (B*R, 4*nreg),
    probs (B*R, ncls), image b's rows [b*R, (b+1)*R); out = (dets (B, ncls, R, 5) f64, counts (B, ncls) i32, keep_idx (B, ncls, R) i32)."""
    rois, deltas, probs = _f32c(rois, "rois"), _f32c(deltas, "deltas"), _f32c(probs, "probs")
    R, ncls = probs.shape[0] // B, probs.shape[1]
    nreg = deltas.shape[1] // 4
    dets, counts, keep_idx = out
    if tuple(dets.shape) != (B, ncls, R, 5) or tuple(counts.shape) != (B, ncls) or not (dets.is_contiguous() and counts.is_contiguous() and
                                                                                       keep_idx.is_contiguous()):
        raise LsfaError("det_postprocess_batch: out must be contiguous (B, ncls, R, 5) / (B, ncls) / (B, ncls, R) buffers")
    if LAB_SKIP_DET:
        return dets, counts, keep_idx
    _check(lib().lsfa_det_postprocess_batch(_ptr(rois), _ptr(deltas), _ptr(probs), _ci(B), _ci(R), _ci(ncls), _ci(nreg),
                                            _ci(int(class_agnostic)), _cd(im_h), _cd(im_w), _cd(scale), _cd(score_thresh),
                                            _cd(nms_thresh), _ci(max_per_image), _ptr(dets), _ptr(counts), _ptr(keep_idx),
                                            None, ctypes.c_size_t(0), _stream()), "lsfa_det_postprocess_batch")
    return dets, counts, keep_idx


@_on_tensor_device
def deform_im2col(data, offset, kh, kw, pad, stride, dilate, deform_groups, out=None):
    data, offset = _f32c(data, "data"), _f32c(offset, "offset")
    N, C, H, W = data.shape
    Ho, Wo = offset.shape[2], offset.shape[3]
    if out is None:
        out = torch.empty((N, C * kh * kw, Ho * Wo), dtype=torch.float32, device=data.device)
    _check(lib().lsfa_deform_im2col(_ptr(data), _ptr(offset), _ci(N), _ci(C), _ci(H), _ci(W), _ci(kh), _ci(kw),
                                    _ci(pad), _ci(stride), _ci(dilate), _ci(deform_groups), _ci(Ho), _ci(Wo),
                                    _ptr(out), _stream()), "lsfa_deform_im2col")
    return out


@_on_tensor_device
def deform_im2col_cl(data, offset, kh, kw, pad, stride, dilate, deform_groups, out=None):
    """data (N,H,W,C), offset (N,Ho,Wo,L) contiguous with L >= 2*kh*kw*dg (channels past those are padding)
    -> col (N, Ho*Wo, kh*kw*C), k = tap*C + c."""
    data, offset = _f32c(data, "data"), _f32c(offset, "offset")
    N, H, W, C = data.shape
    Ho, Wo = offset.shape[1], offset.shape[2]
    if offset.shape[3] < 2 * kh * kw * deform_groups:
        raise LsfaError("deform_im2col_cl: offset has %d channels, expected %d" % (offset.shape[3], 2 * kh * kw * deform_groups))
    if out is None:
        out = torch.empty((N, Ho * Wo, kh * kw * C), dtype=torch.float32, device=data.device)
    _check(lib().lsfa_deform_im2col_cl_ld(_ptr(data), _ptr(offset), _ci(offset.shape[3]), _ci(N), _ci(C), _ci(H), _ci(W),
                                          _ci(kh), _ci(kw), _ci(pad), _ci(stride), _ci(dilate), _ci(deform_groups), _ci(Ho),
                                          _ci(Wo), _ptr(out), _stream()), "lsfa_deform_im2col_cl_ld")
    return out


@_on_tensor_device
def scale_shift_relu(x, scale, shift, relu=True, out=None):
    x = _f32c(x, "x")
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    if out is None:
        out = torch.empty_like(x)
    _check(lib().lsfa_scale_shift_relu(_ptr(x), _ptr(scale), _ptr(shift), _ci(N), _ci(C), _ci(HW), _ci(int(relu)),
                                       _ptr(out), _stream()), "lsfa_scale_shift_relu")
    return out


@_on_tensor_device
def scale_shift_leaky(x, scale, shift, slope, out=None):
    x = _f32c(x, "x")
    N, C = x.shape[0], x.shape[1]
    HW = x.numel() // (N * C)
    if out is None:
        out = torch.empty_like(x)
    _check(lib().lsfa_scale_shift_leaky(_ptr(x), _ptr(scale), _ptr(shift), _ci(N), _ci(C), _ci(HW), _cf(slope),
                                        _ptr(out), _stream()), "lsfa_scale_shift_leaky")
    return out


@_on_tensor_device
def scale_shift_relu_cl(x, scale, shift, relu=True, out=None):
    """Channels-last form: x (..., C) contiguous with the channel the fastest axis, C % 4 == 0."""
    x = _f32c(x, "x")
    C = x.shape[-1]
    rows = x.numel() // C
    if out is None:
        out = torch.empty_like(x)
    _check(lib().lsfa_scale_shift_relu_cl(_ptr(x), _ptr(scale), _ptr(shift), ctypes.c_longlong(rows), _ci(C),
                                          _ci(int(relu)), _ptr(out), _stream()), "lsfa_scale_shift_relu_cl")
    return out


# algorithmic FLOPs (2*M*N*K, fp32-equivalent) of the own convolutions issued since the last reset: bench.py divides
# them by the event-timed duration of the same launches for its MFMA roofline
_conv_flops = {"count": False, "flops": 0.0, "launches": 0, "flops_three_products": 0.0, "flops_one_product": 0.0, "bytes": 0.0, "by_kernel": {}}


def conv_flops_reset(enable=True):
    _conv_flops.update(count=bool(enable), flops=0.0, launches=0, flops_three_products=0.0, flops_one_product=0.0, bytes=0.0, by_kernel={})


def conv_bytes_read():
    """r5: algorithmic bytes of the counted calls - each operand once: the input channels the call reads, its packed weights, the output,
    the residual, a stored second output (4 bytes per activation, 2 x pieces per weight)"""
    return _conv_flops["bytes"]


def conv_by_kernel():
    """r5: {kernel instantiation as rocprofv3 prints it: {"calls", "gflop", "mbytes"}} of the counted calls (lsfa_conv_plan_query)"""
    return {k: {"calls": v[0], "gflop": round(v[1] / 1e9, 3), "mbytes": round(v[2] / 1e6, 3)} for k, v in _conv_flops["by_kernel"].items()}


def _count_conv_launch(name, flops, nbytes):
    if _conv_flops["count"]:
        _conv_flops["bytes"] += nbytes
        e = _conv_flops["by_kernel"].setdefault(name, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += flops
        e[2] += nbytes


def _plan_kernel_name(d):
    q = (ctypes.c_int * 8)()
    if lib().lsfa_conv_plan_query(ctypes.byref(d), q) != 0:
        return "?"
    kind, nt, st, sp, wv, slices, af, pc = [int(v) for v in q]
    b = lambda v: "true" if v else "false"
    if kind == 2:
        return "conv_split_direct_kernel<%d>" % pc
    if kind == 3:
        return "conv_split3x3_kernel<%d, %d>" % (nt, pc)
    return "conv_ring_kernel<%d, %d, %d, %s, %s, %d>%s" % (nt, pc, st, b(sp), b(af), wv, "" if slices <= 1 else " +split_reduce")


def conv_flops_read():
    return _conv_flops["flops"], _conv_flops["launches"]


def conv_flops_three_products():
    """the part of conv_flops_read()'s FLOPs that ran on the fp16 two-piece form (three matrix instructions per product, not six)"""
    return _conv_flops["flops_three_products"]


def conv_flops_one_product():
    """... and the part that ran one bf16 product per fp32 product (the bf16 mode)"""
    return _conv_flops["flops_one_product"]


def _count_conv(N, Ho, Wo, Cout, Cin, kh, kw, pieces=3, launches=1):
    if _conv_flops["count"]:
        f = 2.0 * N * Ho * Wo * Cout * Cin * kh * kw
        _conv_flops["flops"] += f
        _conv_flops["launches"] += launches
        if pieces == 2:
            _conv_flops["flops_three_products"] += f
        elif pieces == 1:
            _conv_flops["flops_one_product"] += f


def conv_weight_kc(weight):
    """(Cout, Cin, kh, kw) -> (Cout, kh*kw, Cin) contiguous: the layout lsfa_conv_nhwc_fwd reads (K contiguous)."""
    co, ci, kh, kw = weight.shape
    return weight.permute(0, 2, 3, 1).reshape(co, kh * kw, ci).contiguous()


@_on_tensor_device
def conv_nhwc(x, w_kc, bias, kh, kw, stride=1, pad=0, dil=1, relu=False, out=None, residual=None, out2=None, scale2=None,
              shift2=None):
    """x (N, H, W, Cin) contiguous fp32; w_kc from conv_weight_kc; -> (N, Ho, Wo, Cout).
    residual (same shape as the output; may BE `out`): added before the ReLU / store.  out2 + scale2 + shift2: second
    output max(out*scale2[c] + shift2[c], 0) (the next ResNet unit's bn1 + relu1).  Returns out, or (out, out2)."""
    x, w_kc = _f32c(x, "x"), _f32c(w_kc, "w_kc")
    N, H, W, Cin = x.shape
    Cout = w_kc.shape[0]
    if tuple(w_kc.shape) != (Cout, kh * kw, Cin):
        raise LsfaError("conv_nhwc: weight %s does not match (Cout, %d, %d)" % (tuple(w_kc.shape), kh * kw, Cin))
    Ho, Wo = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1, (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    if out is None:
        out = torch.empty((N, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    for name, t in (("out", out), ("residual", residual), ("out2", out2)):
        if t is not None and (t.numel() != N * Ho * Wo * Cout or not t.is_contiguous() or t.dtype != torch.float32):
            raise LsfaError("conv_nhwc: %s must be a contiguous float32 tensor of %d elements" % (name, N * Ho * Wo * Cout))
    _count_conv(N, Ho, Wo, Cout, Cin, kh, kw)
    need = lib().lsfa_conv_nhwc_workspace_bytes(_ci(N), _ci(H), _ci(W), _ci(Cout), _ci(kh), _ci(kw), _ci(stride), _ci(pad), _ci(dil))
    ws = torch.empty(need, dtype=torch.uint8, device=x.device)
    _check(lib().lsfa_conv_nhwc_fused_fwd(_ptr(x), _ci(N), _ci(H), _ci(W), _ci(Cin), _ptr(w_kc), _ptr(bias), _ci(Cout), _ci(kh),
                                          _ci(kw), _ci(stride), _ci(pad), _ci(dil), _ci(int(relu)), _ptr(residual), _ptr(out),
                                          _ptr(out2), _ptr(scale2), _ptr(shift2), _ptr(ws), ctypes.c_size_t(need), _stream()),
           "lsfa_conv_nhwc_fused_fwd")
    return out if out2 is None else (out, out2)


def copy_many(pairs):
    """[(dst, src), ...] (at most 32; same shapes, contiguous, 4-byte dtypes, one device) as ONE launch on the current stream.
    src None (r5): dst is zero-filled - the frame path's amax slots and padded maps are cleared by this kernel, not by a PyTorch fill."""
    pairs = [(d, s) for d, s in pairs if d.numel()]
    if not pairs:
        return
    n = len(pairs)
    if n > 32:
        raise LsfaError("copy_many: at most 32 copies per launch")
    for d, s in pairs:
        if d.element_size() != 4 or not d.is_contiguous() or d.device != pairs[0][0].device:
            raise LsfaError("copy_many: %s: need a 4-byte dtype, contiguous, one device" % (tuple(d.shape),))
        if s is not None and (d.shape != s.shape or d.dtype != s.dtype or not s.is_contiguous() or d.device != s.device):
            raise LsfaError("copy_many: %s <- %s: need equal shapes, 4-byte dtype, contiguous, one device"
                            % (tuple(d.shape), tuple(s.shape)))
    if 'copy' in LAB_SKIP:      # ablation: copies of >= 1 KB into a buffer that has been filled once are dropped (zero fills stay)
        kept = []
        for d, s_ in pairs:
            if s_ is None or d.numel() < 256 or d.data_ptr() not in _lab_filled:
                kept.append((d, s_))
            _lab_filled.add(d.data_ptr())
        pairs = kept
        n = len(pairs)
        if not n:
            return
    dst = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in pairs])
    src = (ctypes.c_void_p * n)(*[(s.data_ptr() if s is not None else None) for _, s in pairs])
    cnt = (ctypes.c_long * n)(*[d.numel() for d, _ in pairs])
    with torch.cuda.device(pairs[0][0].device):
        _check(lib().lsfa_copy_many(_ci(n), dst, src, cnt, _stream()), "lsfa_copy_many")


@_on_tensor_device
def transform_mv_res(motion_vector, res_diff, im_scale, pixel_means=(0.0, 0.0, 0.0), pixel_scale=1.0, rcnn_stride=16, negate_mv=False):
    """lsfa_transform_mv_res: (H, W, 2) motion vectors + (H, W, 3) residual on the device (int32 or float32) -> `motion_vector` (1, 2, h, w),
    `res_diff` (1, 3, h, w) float32 (transform_mv_res, lib/utils/image.py:202-228), one launch.  negate_mv: get_image's
    `motion_vector = - motion_vector` (:54) applied to the source values."""
    mv, res = motion_vector, res_diff
    if mv.dtype != res.dtype or mv.dtype not in (torch.int32, torch.float32):
        raise LsfaError("transform_mv_res: int32 or float32 maps expected, got %s / %s" % (mv.dtype, res.dtype))
    if mv.dim() != 3 or res.dim() != 3 or mv.shape[2] != 2 or res.shape[2] != 3 or mv.shape[:2] != res.shape[:2] or not (mv.is_contiguous() and res.is_contiguous()):
        raise LsfaError("transform_mv_res: contiguous (H, W, 2) and (H, W, 3) expected, got %s / %s" % (tuple(mv.shape), tuple(res.shape)))
    H, W = int(mv.shape[0]), int(mv.shape[1])
    import numpy as _np
    h1, w1 = int(_np.rint(H * float(im_scale))), int(_np.rint(W * float(im_scale)))       # cvRound
    oh, ow = -(-h1 // rcnn_stride), -(-w1 // rcnn_stride)
    out_mv = torch.empty((1, 2, oh, ow), device=mv.device, dtype=torch.float32)
    out_res = torch.empty((1, 3, oh, ow), device=mv.device, dtype=torch.float32)
    means = (ctypes.c_double * 3)(*[float(m) for m in pixel_means])
    _check(lib().lsfa_transform_mv_res(_ptr(mv), _ptr(res), _ci(int(mv.dtype == torch.int32) | (2 if negate_mv else 0)), _ci(H), _ci(W), ctypes.c_double(float(im_scale)), _ci(h1), _ci(w1),
                                       _ci(int(rcnn_stride)), means, ctypes.c_double(float(pixel_scale)), _ptr(out_mv), _ptr(out_res), _ci(oh), _ci(ow),
                                       _stream()), "lsfa_transform_mv_res")
    return out_mv, out_res


@_on_tensor_device
def image_resize_transform(im, im_scale, pixel_means=(0.0, 0.0, 0.0), pixel_scale=1.0, stride=0, u8_fixed_point=False):
    """lsfa_image_resize_transform: decoded frames (N, H, W, 3) or (H, W, 3) BGR, uint8 or float32, on the device -> `data` (N, 3, h, w) float32:
    resize (lib/utils/image.py:266-294) + transform (:296-308) in one launch.  u8_fixed_point (uint8 frames only): interpolate on OpenCV's
    fixed-point uint8 path and subtract the means in float64 - what the reference does to the LAST frame of a video, which it reads with
    cv2.imread (:45); the default treats a uint8 frame like the decoder's (converted to float first, :52)."""
    if im.dim() == 3:
        im = im.unsqueeze(0)
    if im.dtype not in (torch.uint8, torch.float32) or im.dim() != 4 or im.shape[3] != 3 or not im.is_contiguous():
        raise LsfaError("image_resize_transform: (N, H, W, 3) contiguous uint8 or float32 expected, got %s %s" % (tuple(im.shape), im.dtype))
    N, H, W, _ = [int(v) for v in im.shape]
    import numpy as _np
    h1, w1 = int(_np.rint(H * float(im_scale))), int(_np.rint(W * float(im_scale)))       # cvRound
    ph, pw = (-(-h1 // stride) * stride, -(-w1 // stride) * stride) if stride else (h1, w1)
    out = torch.empty((N, 3, ph, pw), device=im.device, dtype=torch.float32)
    means = (ctypes.c_double * 3)(*[float(m) for m in pixel_means])
    if u8_fixed_point and im.dtype != torch.uint8:
        raise LsfaError("image_resize_transform: u8_fixed_point needs uint8 frames")
    _check(lib().lsfa_image_resize_transform(_ptr(im), _ci(2 if u8_fixed_point else int(im.dtype == torch.uint8)), _ci(N), _ci(H), _ci(W), ctypes.c_double(float(im_scale)), _ci(h1), _ci(w1),
                                             _ci(int(stride)), means, ctypes.c_double(float(pixel_scale)), _ptr(out), _ci(ph), _ci(pw), _stream()),
           "lsfa_image_resize_transform")
    return out


@_on_tensor_device
def image_transform_u8(im, pixel_means=(0.0, 0.0, 0.0), pixel_scale=1.0, out=None):
    """lsfa_image_transform_u8: decoded frames (N, H, W, 3) uint8 BGR on the device -> (N, 3, H, W) float32 RGB minus means, times scale
    (transform, lib/utils/image.py:296-308).  pixel_means in B, G, R order (config.network.PIXEL_MEANS)."""
    if im.dtype != torch.uint8 or im.dim() != 4 or im.shape[3] != 3 or not im.is_contiguous():
        raise LsfaError("image_transform_u8: (N, H, W, 3) contiguous uint8 expected, got %s %s" % (tuple(im.shape), im.dtype))
    N, H, W, _ = im.shape
    if out is None:
        out = torch.empty((N, 3, H, W), device=im.device, dtype=torch.float32)
    means = (ctypes.c_double * 3)(*[float(m) for m in pixel_means])
    _check(lib().lsfa_image_transform_u8(_vp(im.data_ptr()), _ci(N), _ci(H), _ci(W), means, ctypes.c_double(pixel_scale), _ptr(out), _stream()),
           "lsfa_image_transform_u8")
    return out


class ImageTable(object):
    """N images (C, H, W) float32, each ANYWHERE on the device, in place of one (N, C, H, W) tensor: what lsfa_avgpool_nchw_tbl and
    lsfa_stem_conv7x7s2_tbl read.  The reference's loader hands over one array per frame (dff_rfcn/core/loader.py:131-141); a batched pass
    over F such frames used to copy them into one static buffer first (lsfa_copy_many: 2 x 65 MB per nine-frame segment, 2.4 % of frames/s,
    profiles/r6/tail_ablation.txt).  The table of N device pointers has a fixed address - a captured graph bakes THAT in - and `set`
    rewrites its entries on the current stream with one tiny launch (lsfa_ptr_table_set); the images must stay alive and unchanged until
    the pass that reads them has run (the table keeps references until the next `set`)."""

    def __init__(self, n, c, h, w, device):
        self.shape = (int(n), int(c), int(h), int(w))
        self.device = torch.device(device)
        self.ptrs = torch.zeros(int(n), dtype=torch.int64, device=self.device)
        self.dtype, self.is_cuda = torch.float32, True
        self.held = None

    def dim(self):
        return 4

    def contiguous(self):
        return self

    def is_contiguous(self):
        return True

    def set(self, images):
        """images: N tensors (C, H, W) (or (1, C, H, W)), float32, contiguous, on this device"""
        n, c, h, w = self.shape
        images = list(images)
        if len(images) != n:
            raise LsfaError("ImageTable.set: %d images for a table of %d" % (len(images), n))
        for t in images:
            if t.dtype != torch.float32 or not t.is_cuda or t.device != self.device or not t.is_contiguous() or t.numel() != c * h * w or \
                    tuple(t.shape[-3:]) != (c, h, w) or t.data_ptr() % 4:
                raise LsfaError("ImageTable.set: every image must be a contiguous float32 (%d, %d, %d) tensor on %s, got %s %s on %s" % (
                    c, h, w, self.device, tuple(t.shape), t.dtype, t.device))
        arr = (ctypes.c_void_p * n)(*[t.data_ptr() for t in images])
        with torch.cuda.device(self.device):
            _check(lib().lsfa_ptr_table_set(_vp(self.ptrs.data_ptr()), _ci(n), arr, _stream()), "lsfa_ptr_table_set")
        self.held = images
        return self

    def materialize(self):
        """the images as one (N, C, H, W) tensor (a copy; for code paths that need a tensor)"""
        return torch.stack([t.reshape(self.shape[1:]) for t in self.held], 0)


@_on_tensor_device
def avgpool_nchw(x, k, out=None):
    """(N, C, H, W) float32 (or an ImageTable of N images) -> (N, C, ceil(H/k), ceil(W/k)): k x k / k average, edge windows clipped
    (lsfa_avgpool_nchw / lsfa_avgpool_nchw_tbl)."""
    tbl = isinstance(x, ImageTable)
    if not tbl:
        x = _f32c(x, "x")
    N, C, H, W = x.shape
    Ho, Wo = -(-H // k), -(-W // k)
    if out is None:
        out = torch.empty((N, C, Ho, Wo), device=x.device, dtype=torch.float32)
    if tbl:
        with torch.cuda.device(x.device):
            _check(lib().lsfa_avgpool_nchw_tbl(_vp(x.ptrs.data_ptr()), _ci(N), _ci(C), _ci(H), _ci(W), _ci(k), _ptr(out), _stream()), "lsfa_avgpool_nchw_tbl")
        return out
    _check(lib().lsfa_avgpool_nchw(_ptr(x), _ci(N), _ci(C), _ci(H), _ci(W), _ci(k), _ptr(out), _stream()), "lsfa_avgpool_nchw")
    return out


@_on_tensor_device
def stem_weight_layout(weight):
    """conv0's (64, 3, 7, 7) weight (bn0 folded) -> the fragments lsfa_stem_conv7x7s2 reads (lsfa_stem_weights: a uint8 tensor of
    lsfa_stem_weight_bytes())."""
    co, ci, kh, kw = weight.shape
    if (co, ci, kh, kw) != (64, 3, 7, 7):
        raise LsfaError("stem_weight_layout: expected a (64, 3, 7, 7) weight, got %s" % (tuple(weight.shape),))
    w_l = weight.float().permute(1, 2, 3, 0).contiguous()
    frag = torch.empty((int(lib().lsfa_stem_weight_bytes()),), dtype=torch.uint8, device=weight.device)
    _check(lib().lsfa_stem_weights(_ptr(w_l), _ptr(frag), _stream()), "lsfa_stem_weights")
    return frag


@_on_tensor_device
def stem_conv(x, w_l, bias, in_scale=None, in_shift=None, out=None, accum=None, act=1, amax_out=None):
    """bn_data + conv0 (7x7, stride 2, pad 3) + bias (+ accum) + activation (0 none, 1 ReLU, 2 LeakyReLU 0.1):
    x (N, 3, H, W) NCHW -> (N, Ho, Wo, 64) channels-last.  amax_out: a zeroed row of amax_slots() for max|out|."""
    tbl = isinstance(x, ImageTable)      # the N images given by a table of device pointers (lsfa_stem_conv7x7s2_tbl)
    if not tbl:
        x = _f32c(x, "x")
    N, C, H, W = x.shape
    if C != 3 or w_l.dtype != torch.uint8 or w_l.numel() != int(lib().lsfa_stem_weight_bytes()) or not w_l.is_contiguous():
        raise LsfaError("stem_conv: x must have 3 channels and w_l must come from stem_weight_layout")
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty((N, Ho, Wo, 64), device=x.device, dtype=torch.float32)
    if accum is not None and (tuple(accum.shape) != tuple(out.shape) or not accum.is_contiguous()):
        raise LsfaError("stem_conv: accum must be a contiguous %s tensor" % (tuple(out.shape),))
    if tbl:
        with torch.cuda.device(x.device):
            _check(lib().lsfa_stem_conv7x7s2_tbl(_vp(x.ptrs.data_ptr()), _ci(N), _ci(H), _ci(W), _ptr(in_scale), _ptr(in_shift), _ptr(w_l), _ptr(bias),
                                                 _ptr(accum), _ci(act), _ptr(out), _ptr(amax_out), _stream()), "lsfa_stem_conv7x7s2_tbl")
        return out
    _check(lib().lsfa_stem_conv7x7s2_ex(_ptr(x), _ci(N), _ci(H), _ci(W), _ptr(in_scale), _ptr(in_shift), _ptr(w_l), _ptr(bias),
                                        _ptr(accum), _ci(act), _ptr(out), _ptr(amax_out), _stream()), "lsfa_stem_conv7x7s2_ex")
    return out


@_on_tensor_device
def head_conv3x3(x, w, bias, cin=None, mul=1.0, out=None, c0=0, nchw=False):
    """lsfa_head_conv3x3: 3x3 / pad 1 convolution to <= 4 channels (FlowNet's flow heads).  x (N,H,W,L) channels-last, w (Cout,3,3,Cin);
    nchw: returns (N,Cout,H,W); else writes channels [c0, c0+Cout) of `out` (N,H,W,Lout) (a fresh (N,H,W,Cout) map by default)."""
    x, w = _f32c(x, "x"), _f32c(w, "w")
    N, H, W, L = x.shape
    Cout, cin = w.shape[0], (w.shape[3] if cin is None else cin)
    if tuple(w.shape[1:3]) != (3, 3) or w.shape[3] != cin or cin > L:
        raise LsfaError("head_conv3x3: weight %s does not fit %d input channels of %d" % (tuple(w.shape), cin, L))
    if out is None:
        out = torch.empty((N, Cout, H, W) if nchw else (N, H, W, Cout), device=x.device, dtype=torch.float32)
    _check(lib().lsfa_head_conv3x3(_ptr(x), _ci(L), _ci(N), _ci(H), _ci(W), _ci(cin), _ptr(w), _ptr(bias), _ci(Cout), _cf(mul), _ptr(out),
                                   _ci(int(nchw)), _ci(out.shape[3] if not nchw else 0), _ci(c0), _stream()), "lsfa_head_conv3x3")
    return out


@_on_tensor_device
def upsample_flow(x, w, bias, out, c0, amax_out=None):
    """lsfa_upsample_flow: Deconvolution(4x4, stride 2) + Crop(offset 1) of a (N,Hi,Wi,C) flow into channels [c0, c0+C) of out (N,Hc,Wc,L)."""
    x, w = _f32c(x, "x"), _f32c(w, "w")
    N, Hi, Wi, C = x.shape
    _check(lib().lsfa_upsample_flow(_ptr(x), _ci(N), _ci(Hi), _ci(Wi), _ci(C), _ptr(w), _ptr(bias), _ci(out.shape[1]), _ci(out.shape[2]),
                                    _ptr(out), _ci(out.shape[3]), _ci(c0), _ptr(amax_out), _stream()), "lsfa_upsample_flow")
    return out


@_on_tensor_device
def nchw_to_nhwc(x, c0=0, c=None, out=None, amax_out=None):
    """channels [c0, c0 + c) of an (N, C, H, W) map -> (N, H, W, c) channels-last (lsfa_nchw_to_nhwc), into `out` when given (a contiguous
    (N, H, W, c) tensor or leading slice of one); amax_out: a zeroed row of amax_slots() that receives max|.| of the copied values."""
    x = _f32c(x, "x")
    N, C, H, W = x.shape
    c = C - c0 if c is None else c
    if out is None:
        out = torch.empty((N, H, W, c), device=x.device, dtype=torch.float32)
    elif tuple(out.shape) != (N, H, W, c) or not out.is_contiguous() or out.dtype != torch.float32:
        raise LsfaError("nchw_to_nhwc: out must be a contiguous float32 %s tensor" % ((N, H, W, c),))
    if 'nhwc' in LAB_SKIP and out.data_ptr() in _lab_filled:
        return out
    _lab_filled.add(out.data_ptr())
    _check(lib().lsfa_nchw_to_nhwc(_ptr(x), _ci(N), _ci(C), _ci(H * W), _ci(c0), _ci(c), _ptr(out), _ptr(amax_out), _stream()),
           "lsfa_nchw_to_nhwc")
    return out


def rfcn_head_ps_ld(ps_map, cell_ld, rois, H, W, ncls, nbox, spatial_scale=0.0625, pooled_size=7, group_size=7):
    """rfcn_head_ps on a position-sensitive map whose cells are `cell_ld` floats apart (N, H, W, cell_ld)."""
    ps_map, rois = _f32c(ps_map, "ps_map"), _f32c(rois, "rois")
    N, R = ps_map.shape[0], rois.shape[0]
    cls_prob = torch.empty((R, ncls), device=rois.device, dtype=torch.float32)
    bbox_pred = torch.empty((R, nbox), device=rois.device, dtype=torch.float32)
    if LAB_SKIP_HEAD:
        return cls_prob, bbox_pred
    with torch.cuda.device(ps_map.device):
        _check(lib().lsfa_rfcn_head_ps_ld_fwd(_ptr(ps_map), _ci(cell_ld), _ptr(rois), _ci(N), _ci(H), _ci(W), _ci(R), _ci(ncls), _ci(nbox),
                                              _cf(spatial_scale), _ci(pooled_size), _ci(group_size), _ptr(cls_prob), None,
                                              _ptr(bbox_pred), _stream()), "lsfa_rfcn_head_ps_ld_fwd")
    return cls_prob, bbox_pred


@_on_tensor_device
def avgpool2_nhwc(x):
    """(N,H,W,C) -> (N,ceil(H/2),ceil(W/2),C), 2x2 / 2 average with clipped edge windows (lsfa_avgpool2_nhwc)."""
    x = _f32c(x, "x")
    N, H, W, C = x.shape
    y = torch.empty((N, (H + 1) // 2, (W + 1) // 2, C), device=x.device, dtype=torch.float32)
    _check(lib().lsfa_avgpool2_nhwc(_ptr(x), _ci(N), _ci(H), _ci(W), _ci(C), _ptr(y), _stream()), "lsfa_avgpool2_nhwc")
    return y


@_on_tensor_device
def maxpool3x3s2_nhwc(x, out=None, scale2=None, shift2=None, amax_out=None):
    """(N, H, W, C) float32 channels-last -> (N, (H-1)//2+1, (W-1)//2+1, C): 3x3, stride 2, pad 1 max pooling.
    With scale2 / shift2 (C): returns (pooled, max(pooled*scale2 + shift2, 0)) — the first unit's bn1 + relu1 in the same launch.
    amax_out: a zeroed row of amax_slots() that receives the maximum of the map the next convolution reads (the second output
    when there is one)."""
    x = _f32c(x, "x")
    N, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    if out is None:
        out = torch.empty((N, Ho, Wo, C), device=x.device, dtype=torch.float32)
    out2 = torch.empty_like(out) if scale2 is not None else None
    _check(lib().lsfa_maxpool3x3s2_nhwc(_ptr(x), _ci(N), _ci(H), _ci(W), _ci(C), _ptr(out), _ptr(out2), _ptr(scale2), _ptr(shift2),
                                        _ptr(amax_out), _stream()), "lsfa_maxpool3x3s2_nhwc")
    return out if out2 is None else (out, out2)


class ConvDesc(ctypes.Structure):
    """struct lsfa_conv_desc (include/lsfa_hip.h)"""
    _fields_ = [("x", ctypes.c_void_p), ("lda", ctypes.c_int), ("N", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("Cin", ctypes.c_int),
                ("wfrag", ctypes.c_void_p), ("pieces", ctypes.c_int), ("w_exp", ctypes.c_int), ("amax_in", ctypes.c_void_p),
                ("bias", ctypes.c_void_p), ("Cout", ctypes.c_int), ("kh", ctypes.c_int), ("kw", ctypes.c_int), ("stride", ctypes.c_int),
                ("pad_h", ctypes.c_int), ("pad_w", ctypes.c_int), ("dil", ctypes.c_int),
                ("act", ctypes.c_int), ("y_nchw", ctypes.c_int), ("residual", ctypes.c_void_p), ("y", ctypes.c_void_p), ("ldy", ctypes.c_int),
                ("y2", ctypes.c_void_p), ("scale2", ctypes.c_void_p), ("shift2", ctypes.c_void_p),
                ("amax_out", ctypes.c_void_p), ("status", ctypes.c_void_p),
                ("Ho", ctypes.c_int), ("Wo", ctypes.c_int), ("out_H", ctypes.c_int), ("out_W", ctypes.c_int), ("out_sy", ctypes.c_int),
                ("out_sx", ctypes.c_int), ("prof_tag", ctypes.c_int), ("x_nchw", ctypes.c_int),
                ("in_scale", ctypes.c_void_p), ("in_shift", ctypes.c_void_p), ("w_scale", ctypes.c_void_p)]


def _dp(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# SplitWeight's default form: three bf16 pieces (exact cut, no scale needed).  The frame path binds its fp32 convolutions with
# pieces=2 (two fp16 pieces + a scale, three products) and its bf16 mode with pieces=1 (lsfa_amd/symbols).
DEFAULT_PIECES = 3
AMAX_SLOTS = 256


class SplitWeight(object):
    """A convolution weight cut into `pieces` pieces per value and laid out in MFMA fragment order (lsfa_conv_weights); made once per
    layer at bind time.  pieces = 2: the values are scaled by 2^w_exp so that the largest lands in [2^13, 2^14)."""

    def __init__(self, weight, real_cout=None, real_cin=None, into=None, pieces=None, per_channel_scale=True):
        """weight: (Cout, Cin, kh, kw) float32 CUDA tensor (the framework's layout).  real_cout / real_cin: the layer's own
        channel counts when `weight` was zero-padded to the kernel's tile sizes (algorithmic FLOPs are counted on those)."""
        w_kc = _f32c(conv_weight_kc(weight), "weight")
        self.pieces = int(DEFAULT_PIECES if pieces is None else pieces)
        self.cout, self.cin, self.kh, self.kw = [int(v) for v in weight.shape]
        self.real_cout, self.real_cin = int(real_cout or self.cout), int(real_cin or self.cin)
        if self.pieces == 0:
            # r5, the exact-fp32 evaluation (a REFERENCE mode for tests and one bench line, LSFA_CONV_PIECES=0): the weight stays fp32 in the
            # (Cout, taps, Cin) layout lsfa_conv_nhwc_fused_fwd reads; conv_split / conv_split_view dispatch on pieces == 0 (_conv_exact)
            if self.cin % 32 or self.cout % 64:
                raise LsfaError("SplitWeight: Cin=%d must be a multiple of 32, Cout=%d of 64" % (self.cin, self.cout))
            self.w_exp, self.frag, self.w_kc = 0, None, w_kc
            return
        need = lib().lsfa_conv_weight_bytes(_ci(self.cout), _ci(self.kh), _ci(self.kw), _ci(self.cin), _ci(self.pieces))
        if need == 0:
            raise LsfaError("SplitWeight: Cin=%d must be a multiple of 32, Cout=%d of 64, pieces=%d one of 1, 2, 3" % (self.cin, self.cout, self.pieces))
        self.w_exp, self.w_scale = 0, None
        self.frag = torch.empty(need, dtype=torch.uint8, device=weight.device) if into is None else into
        if self.frag.numel() != need or not self.frag.is_contiguous():
            raise LsfaError("SplitWeight: `into` must be a contiguous uint8 tensor of %d bytes" % need)
        if self.pieces == 2 and per_channel_scale:
            # r5: one power of two per OUTPUT channel, from the channel's own maximum (a BatchNorm folded into trained weights spreads the
            # channels over many octaves: with one scale per tensor the small channels' lo pieces go subnormal).  An all-zero (padding)
            # channel gets exponent 0.
            amax_c = weight.detach().abs().reshape(self.cout, -1).amax(1).double()
            ok = torch.isfinite(amax_c) & (amax_c > 0)
            e = torch.where(ok, 13 - torch.floor(torch.log2(torch.where(ok, amax_c, torch.ones_like(amax_c)))), torch.zeros_like(amax_c))
            e = e.clamp(-100, 100).to(torch.int32).contiguous()
            self.w_exp_pc = e
            self.w_scale = torch.ldexp(torch.ones(self.cout, dtype=torch.float32, device=weight.device), -e).contiguous()
            with torch.cuda.device(weight.device):
                _check(lib().lsfa_conv_weights_pc(_ptr(w_kc), _ci(self.cout), _ci(self.kh), _ci(self.kw), _ci(self.cin), _ptr(e), _ptr(self.frag),
                                                  _stream()), "lsfa_conv_weights_pc")
            return
        if self.pieces == 2:
            amax = float(weight.abs().max().item())              # bind time: a host synchronisation is fine here
            self.w_exp = 0 if not (amax > 0 and math.isfinite(amax)) else 13 - int(math.floor(math.log2(amax)))
            self.w_exp = max(-100, min(100, self.w_exp))
        with torch.cuda.device(weight.device):
            _check(lib().lsfa_conv_weights(_ptr(w_kc), _ci(self.cout), _ci(self.kh), _ci(self.kw), _ci(self.cin), _ci(self.pieces),
                                           _ci(self.w_exp), _ptr(self.frag), _stream()), "lsfa_conv_weights")


def SplitWeightH(weight, **kw):
    """the fp16 two-piece form by name (r3's class)"""
    return SplitWeight(weight, pieces=2, **kw)


@_on_tensor_device
def amax_partial(x, out=None):
    """lsfa_amax_partial: 256 partial maxima of |x| (float32, contiguous, numel % 4 == 0): an `amax_in` for a map that no convolution
    of this library produced (those leave theirs in `amax_out`)."""
    x = _f32c(x, "x")
    if out is None:
        out = torch.empty(AMAX_SLOTS, device=x.device, dtype=torch.float32)
    _check(lib().lsfa_amax_partial(_ptr(x), ctypes.c_longlong(x.numel()), _ptr(out), _stream()), "lsfa_amax_partial")
    return out


def amax_slots(n, device):
    """n zeroed amax_out slot arrays (n, 256) - uint32 bit patterns of non-negative floats, readable as float32 through .view()
    (cleared by lsfa_copy_many's zero-fill job on the current stream: no PyTorch kernel in a captured frame section)"""
    t = torch.empty((n, AMAX_SLOTS), dtype=torch.int32, device=device)
    if t.is_cuda:
        with torch.cuda.device(t.device):
            copy_many([(t, None)])
    else:
        t.zero_()
    return t


def zeros_f32(shape, device):
    """torch.zeros(shape, float32) cleared by lsfa_copy_many's zero-fill job (see amax_slots)"""
    t = torch.empty(shape, dtype=torch.float32, device=device)
    if t.is_cuda:
        with torch.cuda.device(t.device):
            copy_many([(t, None)])
    else:
        t.zero_()
    return t


def new_status(device):
    return torch.zeros(4, dtype=torch.int32, device=device)


@_on_tensor_device
def check_status(status):
    """lsfa_status_check: raises LsfaError if a convolution raised the status word (non-finite output: the fp16 form's scale was an
    under-estimate, or the input held inf / NaN).  Synchronises the current stream."""
    if LAB_SKIP:      # an ablation run's maps are garbage by construction
        return
    _check(lib().lsfa_status_check(_ptr(status), _stream()), "lsfa_status_check")


def _conv_launch(who, x, lda, N, H, W, cin, sw, bias, stride, pad_h, pad_w, dil, act, nchw, residual, y_ptr, ldy, out2, scale2, shift2,
                 amax_in, amax_out, status, grid, view, prof_tag, device, x_nchw=False, in_scale=None, in_shift=None, x_offset=0):
    if sw.pieces == 2 and amax_in is None:
        raise LsfaError("%s: a two-piece (fp16) weight needs amax_in" % who)
    d = ConvDesc()
    d.x, d.lda, d.N, d.H, d.W, d.Cin = x.data_ptr() + x_offset, lda, N, H, W, cin
    d.wfrag, d.pieces, d.w_exp, d.amax_in = sw.frag.data_ptr(), sw.pieces, sw.w_exp, (amax_in.data_ptr() if amax_in is not None else None)
    d.bias, d.Cout, d.kh, d.kw, d.stride, d.pad_h, d.pad_w, d.dil = (bias.data_ptr() if bias is not None else None), sw.cout, sw.kh, sw.kw, stride, pad_h, pad_w, dil
    d.act, d.y_nchw, d.residual, d.y, d.ldy = int(act), int(nchw), (residual.data_ptr() if residual is not None else None), y_ptr, ldy
    d.y2, d.scale2, d.shift2 = (out2.data_ptr() if out2 is not None else None), (scale2.data_ptr() if scale2 is not None else None), (shift2.data_ptr() if shift2 is not None else None)
    d.amax_out, d.status = (amax_out.data_ptr() if amax_out is not None else None), (status.data_ptr() if status is not None else None)
    d.Ho, d.Wo = grid
    d.out_H, d.out_W, d.out_sy, d.out_sx = view
    d.prof_tag = prof_tag
    d.x_nchw = int(x_nchw)
    d.in_scale, d.in_shift = (in_scale.data_ptr() if in_scale is not None else None), (in_shift.data_ptr() if in_shift is not None else None)
    w_scale = getattr(sw, 'w_scale', None)
    d.w_scale = w_scale.data_ptr() if w_scale is not None else None
    need = lib().lsfa_conv_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=device)
    if _conv_flops["count"]:
        Ho = grid[0] if grid[0] > 0 else (H + 2 * pad_h - dil * (sw.kh - 1) - 1) // stride + 1
        Wo = grid[1] if grid[1] > 0 else (W + 2 * pad_w - dil * (sw.kw - 1) - 1) // stride + 1
        outs = 1 + (1 if residual is not None else 0) + (1 if out2 is not None else 0)
        nbytes = 4.0 * N * H * W * cin + 2.0 * sw.pieces * sw.cout * sw.kh * sw.kw * cin + 4.0 * outs * N * Ho * Wo * sw.cout
        _count_conv_launch(_plan_kernel_name(d), 2.0 * N * Ho * Wo * sw.real_cout * sw.real_cin * sw.kh * sw.kw, nbytes)
    _check(lib().lsfa_conv_fwd(ctypes.byref(d), _ptr(ws), ctypes.c_size_t(need), _stream()), who)


def _conv_exact(x, sw, bias, stride, pad, dil, act, out, residual, out2, scale2, shift2, nchw, x_nchw, in_scale, in_shift):
    """conv_split's contract on the EXACT fp32 matrix instructions (lsfa_conv_nhwc_fused_fwd: v_mfma_f32_32x32x2_f32, every product an fp32
    product, bitwise an fmaf chain) - a reference evaluation, not a fast path: what the channels-last kernel does not take (NCHW in / out,
    the input's bn + ReLU at the cut, LeakyReLU, a second output that is only measured) is done by separate elementwise steps with the
    same fp32 operations in the same order as the split kernels' epilogues."""
    if x_nchw:
        x = x[:, :sw.cin].permute(0, 2, 3, 1).contiguous()
    if in_scale is not None:
        x = torch.relu(x * in_scale + in_shift)                       # max(v * scale + shift, 0): two roundings, like affine_relu4
    if nchw:
        res_cl = residual.permute(0, 2, 3, 1).contiguous() if residual is not None else None
        y = conv_nhwc(x, sw.w_kc, bias, sw.kh, sw.kw, stride, pad, dil, relu=(act == 1), residual=res_cl)
        if act == 2:
            y = torch.where(y > 0, y, y * 0.1)
        y = y.permute(0, 3, 1, 2).contiguous()
        if out is not None:
            out.copy_(y)
            y = out
        if out2 is not None:
            out2.copy_(torch.relu(y * scale2.view(1, -1, 1, 1) + shift2.view(1, -1, 1, 1)))
            return y, out2
        return y
    want2 = out2 is not None
    if act == 2:
        y = conv_nhwc(x, sw.w_kc, bias, sw.kh, sw.kw, stride, pad, dil, relu=False, out=None, residual=residual)
        y = torch.where(y > 0, y, y * 0.1)
        if out is not None:
            out.copy_(y)
            y = out
        return y
    r = conv_nhwc(x, sw.w_kc, bias, sw.kh, sw.kw, stride, pad, dil, relu=(act == 1), out=out, residual=residual,
                  out2=out2 if want2 else None, scale2=scale2 if want2 else None, shift2=shift2 if want2 else None)
    return r


@_on_tensor_device
def conv_split(x, sw, bias=None, stride=1, pad=0, dil=1, relu=False, out=None, residual=None, out2=None, scale2=None,
               shift2=None, nchw=False, act=None, amax_in=None, amax_out=None, status=None, x_nchw=False, in_scale=None, in_shift=None):
    """lsfa_conv_fwd: convolution with fp32 in / out and split operands on the matrix pipe; sw: SplitWeight (its `pieces` picks the form).
    x (N, H, W, Cin) contiguous fp32 -> (N, Ho, Wo, Cout), or (N, Cout, Ho, Wo) with nchw=True (then residual / out2 are NCHW too).
    residual (same shape as the output; may BE `out`): added before the activation / store.  out2 + scale2 + shift2: second output
    max(out*scale2[c] + shift2[c], 0) (the next ResNet unit's bn1 + relu1).  act: 0 none, 1 ReLU, 2 LeakyReLU(0.1) (relu=True = act 1).
    amax_in: 256 partial maxima (float32 or int32 bit patterns) bounding |x| - required for two-piece weights; amax_out: 256 int32 slots that
    receive max|out2| (or max|out|); status: the int32 status word.  scale2 + shift2 WITHOUT out2: the second output is not stored, only its
    maximum published (amax_out) - for a consumer that applies the same affine + ReLU itself through in_scale / in_shift (Cin floats each:
    the input becomes max(x*in_scale[k] + in_shift[k], 0) where it is cut; 1x1, pad 0, two-piece or one-piece weights; amax_in must bound
    the ACTIVATED input).  Returns out, or (out, out2)."""
    x = _f32c(x, "x")
    if x_nchw:
        # the input is channels [0, sw.cin) of an NCHW map (N, Ctot, H, W): 1x1 / stride 1 / no padding, small weights (the RPN head)
        N, Ctot, H, W = x.shape
        Cin = sw.cin
        if Ctot < Cin or (sw.kh, sw.kw, stride, pad, dil) != (1, 1, 1, 0, 1):
            raise LsfaError("conv_split: x_nchw needs a 1x1 / stride 1 / pad 0 convolution on at least %d channels" % Cin)
    else:
        N, H, W, Cin = x.shape
        Ctot = Cin
    if Cin != sw.cin:
        raise LsfaError("conv_split: input has %d channels, the weight %d" % (Cin, sw.cin))
    Cout, kh, kw = sw.cout, sw.kh, sw.kw
    Ho, Wo = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1, (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    if out is None:
        out = torch.empty((N, Cout, Ho, Wo) if nchw else (N, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    for name, t in (("out", out), ("residual", residual), ("out2", out2)):
        if t is not None and (t.numel() != N * Ho * Wo * Cout or not t.is_contiguous() or t.dtype != torch.float32):
            raise LsfaError("conv_split: %s must be a contiguous float32 tensor of %d elements" % (name, N * Ho * Wo * Cout))
    if sw.pieces == 0:
        return _conv_exact(x, sw, bias, stride, pad, dil, (1 if relu else 0) if act is None else act, out, residual, out2, scale2, shift2, nchw,
                           x_nchw, in_scale, in_shift)
    if amax_in is None and sw.pieces == 2:
        amax_in = amax_partial(x)
    _count_conv(N, Ho, Wo, sw.real_cout, sw.real_cin, kh, kw, sw.pieces)
    _conv_launch("lsfa_conv_fwd", x, Ctot, N, H, W, Cin, sw, bias, stride, pad, pad, dil, (1 if relu else 0) if act is None else act, nchw,
                 residual, out.data_ptr(), Cout, out2, scale2, shift2, amax_in, amax_out, status, (0, 0), (0, 0, 0, 0), 0, x.device,
                 x_nchw=x_nchw, in_scale=in_scale, in_shift=in_shift)
    return out if out2 is None else (out, out2)


def conv_split_h(x, swh, bias=None, stride=1, pad=0, dil=1, act=0, out=None, nchw=False, amax=None, amax_out=None, status=None):
    """conv_split with r3's argument names for the fp16 two-piece form (amax: the partial maxima of x or of a map that bounds |x|)"""
    return conv_split(x, swh, bias, stride, pad, dil, out=out, nchw=nchw, act=act, amax_in=amax, amax_out=amax_out, status=status)


@_on_tensor_device
def conv_split_view(x, sw, bias, out, stride=1, pad=(0, 0), dil=1, act=0, cin=None, c0=0, grid=None, place=None, amax_in=None,
                    amax_out=None, status=None, cin0=0):
    """lsfa_conv_fwd between VIEWS of wider channels-last maps.
    x    (N, H, W, L) contiguous fp32, the convolution reads channels [cin0, cin0 + cin) of it (cin = sw.cin by default, L >= cin0 + cin;
         cin0 a multiple of 4: the operand pointer stays 16-byte aligned);
    out  (N, Hout, Wout, Lout) contiguous fp32: the result goes to channels [c0, c0 + sw.cout) of it;
    grid (Ho, Wo): the output grid of this launch when smaller than the convolution's; place (y0, x0, sy, sx): output pixel
         (oy, ox) is written to out[:, y0 + oy*sy, x0 + ox*sx] (default: out[:, oy, ox]) - a phase of a transposed convolution;
    pad  (pad_h, pad_w); act 0 none / 1 ReLU / 2 LeakyReLU(0.1)."""
    x = _f32c(x, "x")
    if not (out.is_contiguous() and out.dtype == torch.float32 and out.dim() == 4):
        raise LsfaError("conv_split_view: out must be a contiguous float32 (N, H, W, C) tensor")
    N, H, W, L = x.shape
    cin = sw.cin if cin is None else cin
    if cin != sw.cin or L < cin0 + cin or cin0 < 0 or cin0 % 4:
        raise LsfaError("conv_split_view: the weight has %d input channels, x offers [%d, %d) (of %d)" % (sw.cin, cin0, cin0 + cin, L))
    if c0 < 0 or c0 + sw.cout > out.shape[3] or out.shape[0] != N:
        raise LsfaError("conv_split_view: channels [%d, %d) do not fit out %s" % (c0, c0 + sw.cout, tuple(out.shape)))
    kh, kw = sw.kh, sw.kw
    Hc, Wc = (H + 2 * pad[0] - dil * (kh - 1) - 1) // stride + 1, (W + 2 * pad[1] - dil * (kw - 1) - 1) // stride + 1
    Ho, Wo = grid if grid is not None else (Hc, Wc)
    y0, x0, sy, sx = place if place is not None else (0, 0, 1, 1)
    Hout, Wout, Lout = out.shape[1], out.shape[2], out.shape[3]
    if y0 + (Ho - 1) * sy >= Hout or x0 + (Wo - 1) * sx >= Wout:
        raise LsfaError("conv_split_view: a %dx%d grid placed at (%d,%d) step (%d,%d) leaves out %s" % (Ho, Wo, y0, x0, sy, sx, tuple(out.shape)))
    if sw.pieces == 0:      # the exact-fp32 reference mode (_conv_exact): plain output grids only (the transposed convolutions keep three bf16 pieces)
        if grid is not None or place is not None or (Ho, Wo) != (Hout, Wout):
            raise LsfaError("conv_split_view: the exact-fp32 mode takes whole output grids only")
        xin = x if (L == cin and cin0 == 0) else x[..., cin0:cin0 + cin].contiguous()
        y = _conv_exact(xin, sw, bias, stride, pad[0], dil, act, None, None, None, None, None, False, False, None, None) if pad[0] == pad[1] else None
        if y is None:
            raise LsfaError("conv_split_view: the exact-fp32 mode needs pad_h == pad_w")
        out[..., c0:c0 + sw.cout] = y
        return out
    if amax_in is None and sw.pieces == 2:
        amax_in = amax_partial(x)          # the whole (N, H, W, L) map bounds its leading channels
    first = out.data_ptr() + 4 * ((y0 * Wout + x0) * Lout + c0)
    view = place is not None or (Ho, Wo) != (Hout, Wout)
    _count_conv(N, Ho, Wo, sw.real_cout, sw.real_cin, kh, kw, sw.pieces)
    _conv_launch("lsfa_conv_fwd (view)", x, L, N, H, W, cin, sw, bias, stride, pad[0], pad[1], dil, act, False, None, first, Lout, None, None,
                 None, amax_in, amax_out, status, (Ho, Wo), (Hout if view else 0, Wout, sy, sx), 1, x.device, x_offset=4 * cin0)
    return out


class PhaseWeights(list):
    """The four phase weights of a Deconvolution(4x4, stride 2) as SplitWeight views of ONE allocation (`frag4`)."""


def deconv_phase_weights(wt, cin_pad=None, pieces=None):
    """wt: MXNet Deconvolution weight (Cin, Cout, 4, 4) -> PhaseWeights for lsfa_deconv4x4s2_crop_fwd (input channels zero-padded to
    cin_pad).  Output row 2m + py of the cropped map reads input rows (m - 1, m) through taps ky = (3, 1) when py = 0 and rows
    (m, m + 1) through ky = (2, 0) when py = 1; columns alike.  With two pieces all four phases share ONE weight scale (w_exp)."""
    cin, cout = int(wt.shape[0]), int(wt.shape[1])
    cpad = int(cin_pad or cin)
    pieces = int(DEFAULT_PIECES if pieces is None else pieces)
    if pieces == 0:
        pieces = 3      # exact-fp32 reference mode: the four-phase launch exists on the split kernels only; three bf16 pieces cut the operands exactly
    per = lib().lsfa_conv_weight_bytes(_ci(cout), _ci(2), _ci(2), _ci(cpad), _ci(pieces))
    if per == 0:
        raise LsfaError("deconv_phase_weights: Cin=%d must be a multiple of 32 and Cout=%d of 64" % (cpad, cout))
    frag4 = torch.empty(4 * per, dtype=torch.uint8, device=wt.device)
    out = PhaseWeights()
    # the four phases together hold every tap of wt exactly once: its maximum is each phase's bound
    amax = float(wt.abs().max().item())
    w_exp = 0 if (pieces != 2 or not (amax > 0 and math.isfinite(amax))) else max(-100, min(100, 13 - int(math.floor(math.log2(amax)))))
    for py in (0, 1):
        for px in (0, 1):
            kys, kxs = ((3, 1) if py == 0 else (2, 0)), ((3, 1) if px == 0 else (2, 0))
            wp = torch.zeros((cout, cpad, 2, 2), device=wt.device, dtype=torch.float32)
            wp[:, :cin] = wt[:, :, kys, :][:, :, :, kxs].permute(1, 0, 2, 3)
            sw = SplitWeight.__new__(SplitWeight)
            sw.pieces, sw.w_exp = pieces, w_exp
            sw.cout, sw.cin, sw.kh, sw.kw = cout, cpad, 2, 2
            sw.real_cout, sw.real_cin = cout, cin
            sw.frag = frag4[(py * 2 + px) * per:(py * 2 + px + 1) * per]
            with torch.cuda.device(wt.device):
                _check(lib().lsfa_conv_weights(_ptr(_f32c(conv_weight_kc(wp), "weight")), _ci(cout), _ci(2), _ci(2), _ci(cpad), _ci(pieces),
                                               _ci(w_exp), _ptr(sw.frag), _stream()), "lsfa_conv_weights")
            out.append(sw)
    out.frag4, out.pieces, out.w_exp = frag4, pieces, w_exp
    return out


@_on_tensor_device
def deconv4x4s2_crop(x, sw4, bias, out, c0=0, act=0, amax_in=None, amax_out=None, status=None):
    """lsfa_deconv4x4s2_crop_fwd: Deconvolution(4x4, stride 2) + Crop(offset 1) + bias + activation as one launch.
    x (N, Hi, Wi, L) channels-last (the weights' Cin channels read, L >= Cin); sw4: deconv_phase_weights(...); the result fills
    channels [c0, c0 + Cout) of out (N, Hc, Wc, Lout)."""
    x = _f32c(x, "x")
    N, Hi, Wi, L = x.shape
    cin, cout = sw4[0].cin, sw4[0].cout
    if len(sw4) != 4 or any((s.cin, s.cout, s.kh, s.kw) != (cin, cout, 2, 2) for s in sw4) or cin > L:
        raise LsfaError("deconv4x4s2_crop: four (Cout, Cin, 2, 2) phase weights with Cin <= %d expected" % L)
    if not (out.is_contiguous() and out.dtype == torch.float32 and out.dim() == 4) or c0 + cout > out.shape[3] or out.shape[0] != N:
        raise LsfaError("deconv4x4s2_crop: bad out %s for channels [%d, %d)" % (tuple(out.shape), c0, c0 + cout))
    Hc, Wc, Lout = out.shape[1], out.shape[2], out.shape[3]
    frags = getattr(sw4, 'frag4', None)
    if frags is None:
        raise LsfaError("deconv4x4s2_crop: the phase weights must come from hip.deconv_phase_weights (one allocation)")
    if amax_in is None and sw4.pieces == 2:
        amax_in = amax_partial(x)
    need = lib().lsfa_deconv4x4s2_crop_workspace_bytes(_ci(N), _ci(Hi), _ci(Wi), _ci(cin), _ci(cout), _ci(Hc), _ci(Wc), _ci(sw4.pieces))
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=x.device)
    for i, s in enumerate(sw4):      # four phases, ONE launch
        _count_conv(N, (Hc + 1) // 2, (Wc + 1) // 2, s.real_cout, s.real_cin, 2, 2, sw4.pieces, launches=1 if i == 0 else 0)
    if _conv_flops["count"]:
        _count_conv_launch("conv_ring_kernel (the four phases of a stride-2 transposed convolution)",
                           sum(2.0 * N * ((Hc + 1) // 2) * ((Wc + 1) // 2) * s.real_cout * s.real_cin * 4 for s in sw4),
                           4.0 * N * Hi * Wi * cin + 4 * 2.0 * sw4.pieces * cout * 4 * cin + 4.0 * N * Hc * Wc * cout)
    _check(lib().lsfa_deconv4x4s2_crop_fwd(_ptr(x), _ci(L), _ci(N), _ci(Hi), _ci(Wi), _ci(cin), _ptr(frags), _ci(sw4.pieces), _ci(sw4.w_exp),
                                           _ptr(amax_in), _ptr(bias), _ci(cout), _ci(act), _vp(out.data_ptr() + 4 * c0), _ci(Lout), _ci(Hc), _ci(Wc),
                                           _ptr(amax_out), _ptr(status), _ptr(ws), ctypes.c_size_t(need), _stream()), "lsfa_deconv4x4s2_crop_fwd")
    return out


def conv_plan_override(kernel=0, nt=0, st=0, slices=0):
    """lsfa_conv_plan_override (lab): force the ring kernel (kernel=1), its tile width / ring depth / K slices; zeros = the plan decides"""
    _check(lib().lsfa_conv_plan_override(_ci(kernel), _ci(nt), _ci(st), _ci(slices)), "lsfa_conv_plan_override")


def conv_order_override(tile_order=-1, k_order=-1):
    """lsfa_conv_order_override (lab / tests): workgroup -> tile numbering (0 pixel tile fastest, 1 channel tile fastest) and the ring kernel's walk
    of K (0 tap by tap, 1 channel chunk by channel chunk); -1 = the default"""
    _check(lib().lsfa_conv_order_override(_ci(tile_order), _ci(k_order)), "lsfa_conv_order_override")


class MotionVectorAccumulator(object):
    """Accumulated compressed-domain motion vectors of one GOP on the device
    (coviar_data_loader.c:71-177 with accumulate = 1; see lsfa_mv_* in include/lsfa_hip.h).

        acc = MotionVectorAccumulator(width, height, device)     # identity: the GOP's I-frame
        acc.add_frame(mvs)          # one P-frame's blocks, (n,7) int32 {source,w,h,src_x,src_y,dst_x,dst_y}
        acc.motion_vectors()        # (H,W,2) int32, what load(..., representation=MV, accumulate=True) returns
        acc.residual(bgr_cur, bgr_ref)   # (H,W,3) int32
    """

    def __init__(self, width, height, device='cuda:0'):
        self.width, self.height, self.device = int(width), int(height), torch.device(device)
        self._bufs = [torch.empty((height, width, 2), dtype=torch.int32, device=self.device) for _ in range(2)]
        self._ws = torch.empty(lib().lsfa_mv_workspace_bytes(_ci(width), _ci(height)), dtype=torch.uint8, device=self.device)
        self._cur = 0
        self.reset()

    @property
    def accu(self):
        return self._bufs[self._cur]

    def reset(self):
        with torch.cuda.device(self.device):
            _check(lib().lsfa_mv_identity(_ptr(self._bufs[self._cur]), _ci(self.width), _ci(self.height), _stream()),
                   "lsfa_mv_identity")

    def add_frame(self, mvs, max_block_area=None):
        """mvs (n, 7) int32.  max_block_area: an upper bound of w*h over the blocks (256 for 16x16 macroblocks);
        when omitted it is read from `mvs`, which costs a device-to-host synchronisation if mvs is on the device."""
        if mvs.dtype != torch.int32 or mvs.dim() != 2 or mvs.shape[1] != 7:
            raise LsfaError("mvs must be (n, 7) int32, got %s %s" % (tuple(mvs.shape), mvs.dtype))
        n = mvs.shape[0]
        if max_block_area is not None:
            area = int(max_block_area)
        else:
            area = int((mvs[:, 1].clamp(min=0).long() * mvs[:, 2].clamp(min=0).long()).max().item()) if n else 0
        mvs = mvs.to(self.device).contiguous()
        old, new = self._bufs[self._cur], self._bufs[1 - self._cur]
        with torch.cuda.device(self.device):
            _check(lib().lsfa_mv_accumulate(_ptr(mvs), _ci(n), _ci(area), _ptr(old), _ptr(new), _ci(self.width),
                                            _ci(self.height), _ptr(self._ws), ctypes.c_size_t(self._ws.numel()), _stream()),
                   "lsfa_mv_accumulate")
        self._cur = 1 - self._cur

    def motion_vectors(self):
        mv = torch.empty((self.height, self.width, 2), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _check(lib().lsfa_mv_field(_ptr(self.accu), _ci(self.width), _ci(self.height), _ptr(mv), _stream()), "lsfa_mv_field")
        return mv

    def residual(self, bgr_cur, bgr_ref):
        for t in (bgr_cur, bgr_ref):
            if t.dtype != torch.uint8 or tuple(t.shape) != (self.height, self.width, 3):
                raise LsfaError("frames must be (H, W, 3) uint8")
        bgr_cur, bgr_ref = bgr_cur.to(self.device).contiguous(), bgr_ref.to(self.device).contiguous()
        res = torch.empty((self.height, self.width, 3), dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _check(lib().lsfa_mv_residual(_ptr(bgr_cur), _ptr(bgr_ref), _ptr(self.accu), _ci(self.width), _ci(self.height),
                                          _ptr(res), _stream()), "lsfa_mv_residual")
        return res

    def network_inputs(self, bgr_cur, bgr_ref, im_scale, pixel_means=(0.0, 0.0, 0.0), pixel_scale=1.0, rcnn_stride=16):
        """The frame's `motion_vector` (1, 2, h, w) and `res_diff` (1, 3, h, w) as get_image builds them (lib/utils/image.py:52-63): the
        accumulated motion vectors, negated, and the residual through transform_mv_res - three launches, nothing leaves the device."""
        return transform_mv_res(self.motion_vectors(), self.residual(bgr_cur, bgr_ref), im_scale, pixel_means, pixel_scale, rcnn_stride, negate_mv=True)


# ---- live timing -----------------------------------------------------------------------
def prof_enable(on=True, ops=None):
    """on=True times every op; ops=['warp_bilinear', ...] times only those; on=False stops."""
    mask = 0
    if on:
        mask = -1 if ops is None else sum(1 << OP_NAMES.index(o) for o in ops)
    _check(lib().lsfa_prof_enable(_ci(mask)), "lsfa_prof_enable")


def prof_read():
    """{op_name: (total_ms, launches)} since the last read; synchronises the recorded events."""
    n = len(OP_NAMES)
    ms = (ctypes.c_double * n)()
    cnt = (ctypes.c_int * n)()
    _check(lib().lsfa_prof_read(ms, cnt), "lsfa_prof_read")
    return {OP_NAMES[i]: (ms[i], cnt[i]) for i in range(n)}
