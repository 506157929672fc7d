"""Frame / motion-vector / residual preprocessing (a "next" row).

Counterparts of lib/utils/image.py: `resize` (:266-294), `transform` (:296-308) and
`transform_mv_res` (:202-228).  The reference runs them with cv2 on the host inside a prefetch
process; here they are torch ops on whatever device the input lives on (the frame, the motion
vectors and the residual of a decoded GOP can stay in HBM).  cv2.resize(src, None, None, fx, fy,
INTER_LINEAR) on float input [OpenCV 3.2 resize.cpp, un-vendored — PARITY UNPINNED]: dsize =
cvRound(size * f); source coordinate of destination pixel d = (d + 0.5) / f - 0.5 — the scale is
1/f as GIVEN, not src/dst (they differ whenever size*f is not an integer, the usual case for
im_scale) — two taps with edge clamping, no antialiasing, horizontal pass then vertical pass.
This is cv2's FLOAT path, which is what the motion vectors, the residual and a decoder's frame (converted with
.astype(np.float32) first, image.py:52) take.  The LAST frame of a video is read with cv2.imread (image.py:45) and reaches
`resize()` as a uint8 image: cv2's fixed-point path (11-bit coefficients, result rounded to uint8), which
`resize(..., u8_fixed_point=True)` follows (r6; oracle/np_ref.py::cv2_resize_linear_u8, lsfa_image_resize_transform's
is_u8 = 2).  cv2 is absent, so neither path can be pinned.
"""
import numpy as np
import torch
import torch.nn.functional as F


def _cv_round(x):
    return int(np.rint(x))      # cvRound: nearest, ties to even


def _taps(dst_n, src_n, f, device):
    """Source taps of cv2's INTER_LINEAR along one axis: (i0, i1, weight of i1)."""
    # resize.cpp: `fx = (float)((dx+0.5)*scale_x - 0.5); sx = cvFloor(fx); fx -= sx;` with scale_x = 1. / f in double: the
    # position is rounded to float BEFORE the floor and the subtraction
    pos = ((torch.arange(dst_n, dtype=torch.float64, device=device) + 0.5) * (1.0 / float(f)) - 0.5).to(torch.float32)
    i0f = torch.floor(pos)
    a = pos - i0f
    i0 = i0f.to(torch.int64)
    lo, hi = i0 < 0, i0 >= src_n - 1
    i0 = torch.where(lo, torch.zeros_like(i0), torch.where(hi, torch.full_like(i0, src_n - 1), i0))
    a = torch.where(lo | hi, torch.zeros_like(a), a)
    return i0, torch.clamp(i0 + 1, max=src_n - 1), a


def _resize_hwc(x, fx, fy):
    """x: (H, W, C) float tensor -> (cvRound(H*fy), cvRound(W*fx), C), cv2.resize(fx=, fy=, INTER_LINEAR)."""
    h, w, _ = x.shape
    oh, ow = _cv_round(h * fy), _cv_round(w * fx)
    x0, x1, ax = _taps(ow, w, fx, x.device)
    y0, y1, ay = _taps(oh, h, fy, x.device)
    wt = x.dtype                                                    # float32 images: cv2's CV_32F path; float64: CV_64F (double work type, the SAME float coefficients)
    bx, by = (1 - ax).to(wt), (1 - ay).to(wt)                       # `cbuf[0] = 1.f - fx` in float
    ax, ay = ax.to(wt), ay.to(wt)
    hor = x[:, x0] * bx[None, :, None] + x[:, x1] * ax[None, :, None]
    return hor[y0] * by[:, None, None] + hor[y1] * ay[:, None, None]


def _resize_hwc_u8(x, fx, fy):
    """x: (H, W, C) uint8 tensor -> uint8, OpenCV 3.2's fixed-point INTER_LINEAR (oracle/np_ref.py::cv2_resize_linear_u8 states the source)."""
    h, w, _ = x.shape
    oh, ow = _cv_round(h * fy), _cv_round(w * fx)
    x0, x1, ax = _taps(ow, w, fx, x.device)
    y0, y1, ay = _taps(oh, h, fy, x.device)
    q = lambda c: torch.round(c * 2048.0).to(torch.int32)         # saturate_cast<short>(c * 2048): nearest, ties to even (torch.round is that)
    a0, a1, b0, b1 = q(1.0 - ax), q(ax), q(1.0 - ay), q(ay)
    xi = x.to(torch.int32)
    hor = xi[:, x0] * a0[None, :, None] + xi[:, x1] * a1[None, :, None]
    out = (((b0[:, None, None] * (hor[y0] >> 4)) >> 16) + ((b1[:, None, None] * (hor[y1] >> 4)) >> 16) + 2) >> 2
    return out.to(torch.uint8)


def resize(im, target_size, max_size, stride=0, u8_fixed_point=False):
    """im (H, W, C) BGR tensor -> (resized [padded to `stride`], im_scale).  lib/utils/image.py:266-294.  u8_fixed_point: a uint8 image is
    interpolated on cv2's fixed-point path and stays uint8 (the reference's last frame of a video); else in float."""
    im = torch.as_tensor(im)
    h, w = im.shape[0], im.shape[1]
    im_size_min, im_size_max = min(h, w), max(h, w)
    im_scale = float(target_size) / float(im_size_min)
    if np.round(im_scale * im_size_max) > max_size:
        im_scale = float(max_size) / float(im_size_max)
    out = _resize_hwc_u8(im, im_scale, im_scale) if (u8_fixed_point and im.dtype == torch.uint8) else _resize_hwc(im.float(), im_scale, im_scale)
    if stride == 0:
        return out, im_scale
    ph = int(np.ceil(out.shape[0] / float(stride)) * stride)
    pw = int(np.ceil(out.shape[1] / float(stride)) * stride)
    padded = torch.zeros((ph, pw, out.shape[2]), dtype=torch.float64, device=out.device)      # np.zeros(...): a float64 image (image.py:291)
    padded[:out.shape[0], :out.shape[1]] = out
    return padded, im_scale


def transform(im, pixel_means, pixel_scale):
    """(H, W, 3) BGR -> (1, 3, H, W) RGB minus means, times scale.  lib/utils/image.py:296-308, in its precisions: a float image minus the
    (Python float) mean is a float32 subtraction, a uint8 or float64 (padded by `resize`) image's a float64 one; the product with pixel_scale is float64 (np.zeros), rounded to
    float32 where the reference hands the array to the executor."""
    im = torch.as_tensor(im)
    if im.dtype in (torch.uint8, torch.float64):      # (float64: a frame `resize` padded to the stride)
        x = im.double()
        means = torch.as_tensor(np.asarray(pixel_means, dtype=np.float64), device=im.device)
    else:
        x = im.float()
        means = torch.as_tensor(np.asarray(pixel_means, dtype=np.float32), device=im.device)
    t = torch.stack([x[:, :, 2 - i] - means[2 - i] for i in range(3)], 0).unsqueeze(0)
    return (t.double() * float(pixel_scale)).float()


def transform_mv_res(motion_vector, res_diff, im_scale, pixel_means, pixel_scale, rcnn_stride=16):
    """(H, W, 2) motion vectors + (H, W, 3) residual -> (1,2,h,w), (1,3,h,w) stride-16 float32 tensors.
    lib/utils/image.py:202-228, including its in-place channel loop: channel 2 of the residual is computed from the ALREADY REWRITTEN
    channel 0 (:218-219), and its precisions: the first resize runs on float32 images, the padded maps are np.zeros - float64 - so the
    channel loop, the resize by 1 / stride (cv2's CV_64F path: double work type, float coefficients) and the final scale are float64, rounded to
    float32 once (where the reference hands the arrays to the executor).  Maps on a GPU take ONE launch of lsfa_transform_mv_res (r5), maps
    on the host the same arithmetic in torch; both equal oracle/np_ref.py::transform_mv_res bit for bit."""
    mv, res = torch.as_tensor(motion_vector), torch.as_tensor(res_diff)
    if mv.is_cuda:
        from lsfa_amd import hip
        if mv.dtype not in (torch.int32, torch.float32):
            mv, res = mv.float(), res.float()
        return hip.transform_mv_res(mv.contiguous(), res.to(mv.dtype).contiguous(), im_scale, pixel_means, pixel_scale, rcnn_stride)
    mv = _resize_hwc(mv.float(), im_scale, im_scale)
    res = _resize_hwc(res.float(), im_scale, im_scale)
    im_h, im_w = res.shape[0], res.shape[1]
    p_h = int(np.ceil(im_h / float(rcnn_stride)) * rcnn_stride)
    p_w = int(np.ceil(im_w / float(rcnn_stride)) * rcnn_stride)
    pmv = torch.zeros((p_h, p_w, 2), dtype=torch.float64)
    pres = torch.zeros((p_h, p_w, 3), dtype=torch.float64)
    pmv[:im_h, :im_w] = mv
    pres[:im_h, :im_w] = res
    means = [float(m) for m in pixel_means]
    for i in range(3):     # the reference's in-place loop, in the same order
        pres[:, :, i] = (pres[:, :, 2 - i] - means[2 - i]) * float(pixel_scale)
    s = 1.0 / rcnn_stride
    rmv = _resize_hwc(pmv, s, s) * (float(im_scale) * s)
    rres = _resize_hwc(pres, s, s)
    return rmv.permute(2, 0, 1).unsqueeze(0).contiguous().float(), rres.permute(2, 0, 1).unsqueeze(0).contiguous().float()
