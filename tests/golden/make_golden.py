"""Generate tests/golden/ref_golden.npz by IMPORTING the reference's own numpy helpers.

Run in the build container only (needs /root/reference; nothing here travels to
the GPU box except the .npz it writes):

    python tests/golden/make_golden.py

What is captured — inputs and the reference's outputs:
  G1  lib/rpn/generate_anchor.py::generate_anchors(16, [.5,1,2], [8,16,32])
  G2  lib/nms/nms.py::nms on seeded tie-free boxes, N in {50, 300, 2000}, float32 and
      float64 inputs, thresholds 0.3 and 0.7
  G3  lib/bbox/bbox_transform.py::bbox_pred (= nonlinear_pred) + clip_boxes
  G4  lib/bbox/bbox_transform.py::bbox_overlaps_py
  G5  the frame loop of dff_rfcn/core/tester.py:143-152 + :265-281 assembled from the
      reference's bbox_pred/clip_boxes/nms (the loop itself is Python 2 and cannot
      be imported, so its ~15 lines are replayed here around the reference's functions)
  G6  (r5) lib/utils/image.py imported with STUB modules for cv2 and coviar_py2 (both absent from this
      image).  `transform` (:296-308) is pure numpy: its output is the reference's, exactly.
      `transform_mv_res` (:202-228), `resize` (:266-294) and `tensor_vstack` run the reference's own code
      with ONE substitution: cv2.resize is the repo's restatement of OpenCV 3.2's INTER_LINEAR
      (oracle/np_ref.py::cv2_resize_linear) - so the padding, the in-place channel loop (:218-219), the
      scale factors, the transposes and the im_scale rule are pinned to the reference; the
      interpolation arithmetic inside cv2.resize stays a restatement (no OpenCV here).
The three import shims are the ones SURVEY.md §8c lists: xrange, np.float, and
stub modules for the missing Cython sources (bbox, cpu_nms, gpu_nms).
"""
import builtins
import os
import sys
import types

import numpy as np

REF = os.environ.get("LSFA_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_golden.npz")


def import_reference():
    builtins.xrange = range
    if not hasattr(np, "float"):
        np.float = float
    for name in ("cpu_nms", "gpu_nms"):          # compiled from the missing .pyx files
        m = types.ModuleType(name)
        setattr(m, name, None)
        sys.modules[name] = m
    sys.path.insert(0, os.path.join(REF, "lib"))
    import bbox as _bbox_pkg                      # lib/bbox/__init__.py; `from bbox import
    _bbox_pkg.bbox_overlaps_cython = None         # bbox_overlaps_cython` (bbox.pyx is missing)
    from nms.nms import nms
    from rpn.generate_anchor import generate_anchors
    from bbox.bbox_transform import bbox_pred, clip_boxes, bbox_overlaps_py
    return nms, generate_anchors, bbox_pred, clip_boxes, bbox_overlaps_py


def import_reference_image():
    """lib/utils/image.py with cv2 / coviar_py2 stubbed (G6).  cv2.resize -> the oracle's restatement of INTER_LINEAR."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from oracle import np_ref
    cv2 = types.ModuleType("cv2")
    cv2.INTER_LINEAR = 1

    def resize(src, dsize, dst=None, fx=None, fy=None, interpolation=1):
        assert dsize is None and interpolation == 1
        return np_ref.cv2_resize_linear(np.asarray(src), fx, fy)
    cv2.resize = resize
    sys.modules["cv2"] = cv2
    sys.modules["coviar_py2"] = types.ModuleType("coviar_py2")
    from utils import image
    return image


def random_boxes(rs, n, im_w=1000, im_h=600, dtype=np.float32, clustered=True):
    """Tie-free scored boxes with plenty of overlap (clusters around a few centres)."""
    if clustered:
        k = max(3, n // 25)
        cx = rs.uniform(50, im_w - 50, k)[rs.randint(0, k, n)] + rs.normal(0, 12, n)
        cy = rs.uniform(50, im_h - 50, k)[rs.randint(0, k, n)] + rs.normal(0, 12, n)
    else:
        cx, cy = rs.uniform(0, im_w, n), rs.uniform(0, im_h, n)
    w, h = rs.uniform(20, 220, n), rs.uniform(20, 220, n)
    x1 = np.clip(cx - w / 2, 0, im_w - 1)
    y1 = np.clip(cy - h / 2, 0, im_h - 1)
    x2 = np.clip(cx + w / 2, 0, im_w - 1)
    y2 = np.clip(cy + h / 2, 0, im_h - 1)
    scores = rs.permutation(n).astype(np.float64) / n * 0.98 + 0.01 + rs.uniform(0, 1e-3 / n, n)
    assert len(np.unique(scores.astype(dtype))) == n, "scores must be tie-free"
    return np.stack([x1, y1, x2, y2, scores], 1).astype(dtype)


def main():
    nms, generate_anchors, bbox_pred, clip_boxes, bbox_overlaps_py = import_reference()
    g = {}
    # G1
    g["g1_anchors"] = generate_anchors(base_size=16, ratios=[0.5, 1, 2], scales=np.array([8, 16, 32]))
    g["g1_anchors_default"] = generate_anchors()
    # G2
    for n in (50, 300, 2000):
        for dt in (np.float32, np.float64):
            rs = np.random.RandomState(1000 + n)
            dets = random_boxes(rs, n, dtype=dt)
            tag = "g2_n%d_%s" % (n, np.dtype(dt).name)
            g[tag + "_dets"] = dets
            for th in (0.3, 0.7):
                g[tag + "_keep_%02d" % int(th * 10)] = np.asarray(nms(dets, th), dtype=np.int64)
    g["g2_empty_keep"] = np.asarray(nms(np.zeros((0, 5), np.float32), 0.3), dtype=np.int64)
    # G3
    rs = np.random.RandomState(7)
    rois = random_boxes(rs, 300, dtype=np.float32, clustered=False)[:, :4]
    deltas = (rs.normal(0, 0.25, (300, 8))).astype(np.float32)
    pb = bbox_pred(rois, deltas)
    g["g3_rois"], g["g3_deltas"], g["g3_pred"] = rois, deltas, pb.copy()
    g["g3_clip"] = clip_boxes(pb.copy(), (600, 1000))
    g["g3_empty"] = bbox_pred(np.zeros((0, 4), np.float32), np.zeros((0, 8), np.float32))
    # G4
    a = random_boxes(np.random.RandomState(11), 40, dtype=np.float64)[:, :4]
    b = random_boxes(np.random.RandomState(12), 25, dtype=np.float64)[:, :4]
    g["g4_a"], g["g4_b"], g["g4_iou"] = a, b, bbox_overlaps_py(a, b)
    # G5: frame post-processing replayed around the reference's functions
    rs = np.random.RandomState(21)
    R, ncls = 300, 31
    rois5 = np.hstack([np.zeros((R, 1), np.float32), random_boxes(rs, R, dtype=np.float32)[:, :4]])
    deltas = rs.normal(0, 0.15, (R, 8)).astype(np.float32)
    logits = rs.normal(0, 2.0, (R, ncls)).astype(np.float32)
    logits[:, 0] += 1.0
    e = np.exp(logits - logits.max(1, keepdims=True))
    probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
    scale, im_shape, thresh, max_per_image = 1.25, (1, 3, 600, 1000), 1e-4, 300
    pred_boxes = clip_boxes(bbox_pred(rois5[:, 1:], deltas), im_shape[-2:]) / scale     # tester.py:147-152
    all_boxes = [np.zeros((0, 5))] * ncls
    for j in range(1, ncls):                                                             # tester.py:266-272
        indexes = np.where(probs[:, j] > thresh)[0]
        cls_scores = probs[indexes, j, np.newaxis]
        cls_boxes = pred_boxes[indexes, 4:8]
        cls_dets = np.hstack((cls_boxes, cls_scores))
        keep = nms(cls_dets, 0.3)
        all_boxes[j] = cls_dets[keep, :]
    image_scores = np.hstack([all_boxes[j][:, -1] for j in range(1, ncls)])              # tester.py:274-281
    g["g5_n_before_cap"] = np.asarray(len(image_scores))
    if len(image_scores) > max_per_image:
        image_thresh = np.sort(image_scores)[-max_per_image]
        for j in range(1, ncls):
            keep = np.where(all_boxes[j][:, -1] >= image_thresh)[0]
            all_boxes[j] = all_boxes[j][keep, :]
    g["g5_rois"], g["g5_deltas"], g["g5_probs"] = rois5, deltas, probs
    g["g5_scale"] = np.asarray(scale)
    g["g5_pred_boxes"] = pred_boxes
    g["g5_counts"] = np.asarray([len(all_boxes[j]) for j in range(ncls)], dtype=np.int64)
    g["g5_dets"] = np.vstack([all_boxes[j] for j in range(ncls)])
    # G6: lib/utils/image.py
    image = import_reference_image()
    rs = np.random.RandomState(33)
    im = rs.randint(0, 256, (20, 28, 3)).astype(np.uint8)                  # BGR, H*W % 4 == 0
    means, pscale = np.array([103.06, 115.90, 123.15]), 0.0167
    g["g6_im"], g["g6_means"], g["g6_pixel_scale"] = im, means, np.asarray(pscale)
    g["g6_transform"] = image.transform(im, means, pscale)                 # exact: pure numpy
    g["g6_transform_zero_means"] = image.transform(im, np.zeros(3), 1.0)   # the resnet-101 configuration (config.py:171-176)
    # a float32 frame (the decoder's, after get_image's .astype(np.float32)) with list means, as update_network_config leaves them
    # (config.py:177-182): the subtraction is a float32 one
    g["g6_transform_f32_list_means"] = image.transform(im.astype(np.float32) * np.float32(0.731), [103.94, 116.78, 123.68], 0.017)
    mv = rs.randint(-40, 41, (37, 52, 2)).astype(np.int32)                 # odd sizes: padding to the stride, 'full' edges
    res = rs.randint(-128, 128, (37, 52, 3)).astype(np.int32)
    g["g6_mv"], g["g6_res"] = mv, res
    for tag, sc in (("s1", 1.0), ("s16", 1.6)):
        t_mv, t_res = image.transform_mv_res(mv, res, sc, means, pscale)
        g["g6_mv_tensor_" + tag], g["g6_res_tensor_" + tag] = t_mv, t_res
    big = rs.randint(0, 256, (30, 50, 3)).astype(np.uint8)
    g["g6_resize_in"] = big
    r_im, r_scale = image.resize(big, 60, 100, stride=16)                  # short side -> 60, capped by the long side; padded to 16
    g["g6_resize_out"], g["g6_resize_scale"] = r_im, np.asarray(r_scale)
    # the decoder's float32 frame padded to the stride: `resize` returns a float64 image then, and `transform` subtracts in float64
    r_f32, _ = image.resize(big.astype(np.float32), 60, 100, stride=16)
    g["g6_resize_f32_stride16_transform"] = image.transform(r_f32, [103.94, 116.78, 123.68], 0.017)
    r_im2, r_scale2 = image.resize(big, 60, 90, stride=0)                  # the max_size rule takes over
    g["g6_resize_out_capped"], g["g6_resize_scale_capped"] = r_im2, np.asarray(r_scale2)
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(g), "arrays")


if __name__ == "__main__":
    main()
