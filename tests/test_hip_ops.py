"""Parity of every hand-written HIP kernel with the CPU oracle, through the C ABI.

All custom ops share the oracle's floating-point contract (-ffp-contract=off, fmaf
only where written, correctly rounded exp), so the assertion is BIT-EXACT equality
for floats and indices alike, unless a test says otherwise.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import np_ref

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def rand_rois(rs, R, im_w=1000, im_h=600, small=0.1):
    cx, cy = rs.uniform(0, im_w, R), rs.uniform(0, im_h, R)
    w, h = rs.uniform(8, 400, R), rs.uniform(8, 400, R)
    tiny = rs.rand(R) < small
    w[tiny], h[tiny] = rs.uniform(0, 3, tiny.sum()), rs.uniform(0, 3, tiny.sum())
    x1, y1 = np.clip(cx - w / 2, 0, im_w - 1), np.clip(cy - h / 2, 0, im_h - 1)
    x2, y2 = np.clip(cx + w / 2, 0, im_w - 1), np.clip(cy + h / 2, 0, im_h - 1)
    rois = np.stack([np.zeros(R), x1, y1, x2, y2], 1).astype(np.float32)
    # some ROIs whose edges land exactly on integers / half-integers (round() and bin-edge cases)
    rois[::7, 1:] = np.round(rois[::7, 1:])
    rois[3::11, 1:] = np.floor(rois[3::11, 1:]) + 0.5
    rois[5::13, 1:] = (np.round(rois[5::13, 1:] / 16) * 16)
    return rois


# ------------------------------------------------------------------ PSROI ------------
@pytest.mark.parametrize("shape", [(38, 63, 300, 31), (38, 63, 300, 8), (36, 63, 77, 31), (5, 9, 16, 2)])
def test_psroi_pool_bit_exact(hip, shape):
    H, W, R, D = shape
    rs = np.random.RandomState(H * 1000 + R)
    data = rs.randn(1, D * 49, H, W).astype(np.float32)
    rois = rand_rois(rs, R, im_w=W * 16, im_h=H * 16)
    want, want_mc = oracle.psroi_pool(data, rois, 0.0625, D, 7, 7, with_mapping=True)
    got, got_mc = hip.psroi_pool(t(data), t(rois), 0.0625, D, 7, 7, with_mapping=True)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    np.testing.assert_array_equal(got_mc.cpu().numpy(), want_mc)


def test_psroi_pool_batch_index_and_groups(hip):
    rs = np.random.RandomState(3)
    data = rs.randn(2, 3 * 9, 10, 12).astype(np.float32)
    rois = rand_rois(rs, 40, im_w=12 * 8, im_h=10 * 8)
    rois[:, 0] = rs.randint(0, 2, 40)
    want = oracle.psroi_pool(data, rois, 0.125, 3, 3, 3)
    got = hip.psroi_pool(t(data), t(rois), 0.125, 3, 3, 3)
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_psroi_errors(hip):
    data = torch.zeros(1, 10, 4, 4, device=DEV)
    rois = torch.zeros(2, 5, device=DEV)
    with pytest.raises(hip.LsfaError):  # channels != output_dim * group^2 (psroi_pooling-inl.h InferShape)
        hip.psroi_pool(data, rois, 0.0625, 3, 7, 7)
    out = hip.psroi_pool(torch.zeros(1, 49, 4, 4, device=DEV), torch.zeros(0, 5, device=DEV), 0.0625, 1, 7, 7)
    assert out.shape == (0, 1, 7, 7)


@pytest.mark.parametrize("R", [300, 1])
def test_rfcn_head_fused_bit_exact(hip, R):
    rs = np.random.RandomState(11 + R)
    cls_map = rs.randn(1, 31 * 49, 38, 63).astype(np.float32)
    box_map = (0.1 * rs.randn(1, 8 * 49, 38, 63)).astype(np.float32)
    rois = rand_rois(rs, R)
    want_prob, want_score, want_box = oracle.rfcn_head(cls_map, box_map, rois)
    prob, score, box = hip.rfcn_head(t(cls_map), t(box_map), t(rois), want_score=True)
    np.testing.assert_array_equal(score.cpu().numpy(), want_score)
    np.testing.assert_array_equal(box.cpu().numpy(), want_box)
    np.testing.assert_array_equal(prob.cpu().numpy(), want_prob)
    # and the fused path equals the unfused operator + average
    unf = hip.psroi_pool(t(cls_map), t(rois), 0.0625, 31, 7, 7).cpu().numpy()
    np.testing.assert_array_equal(oracle.global_avg(unf), want_score)


def test_rfcn_head_position_sensitive_layout_bit_exact(hip):
    rs = np.random.RandomState(21)
    H, W, R = 38, 63, 300
    cls_map = rs.randn(1, 31 * 49, H, W).astype(np.float32)
    box_map = (0.1 * rs.randn(1, 8 * 49, H, W)).astype(np.float32)
    rois = rand_rois(rs, R)
    want_prob, want_score, want_box = oracle.rfcn_head(cls_map, box_map, rois)
    nchw = np.concatenate([cls_map.reshape(31, 49, H * W), box_map.reshape(8, 49, H * W)], 0)   # (D, 49, HW)
    ps = np.ascontiguousarray(nchw.transpose(2, 1, 0)).reshape(1, H, W, 49, 39)
    prob, score, box = hip.rfcn_head_ps(t(ps), t(rois), 31, 8, want_score=True)
    np.testing.assert_array_equal(score.cpu().numpy(), want_score)
    np.testing.assert_array_equal(box.cpu().numpy(), want_box)
    np.testing.assert_array_equal(prob.cpu().numpy(), want_prob)


# ------------------------------------------------------------------ warp --------------
def smooth_flow(rs, N, H, W, mag):
    f = rs.uniform(-mag, mag, (N, 2, 1, 1)) + 0.3 * rs.randn(N, 2, H, W)
    return f.astype(np.float32)


@pytest.mark.parametrize("shape", [(1, 1024, 38, 63), (1, 64, 36, 63), (2, 32, 16, 16), (1, 16, 7, 5)])
def test_warp_plain_bit_exact_and_grid_sample(hip, shape):
    N, C, H, W = shape
    rs = np.random.RandomState(C + H)
    feat = rs.randn(N, C, H, W).astype(np.float32)
    flow = smooth_flow(rs, N, H, W, 2.5)
    want = oracle.warp_bilinear(feat, flow)
    got = hip.warp_bilinear(t(feat), t(flow)).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    # independent cross-check of the un-pinned MXNet semantics: grid_sample(align_corners=True, zeros)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    gx = (xs[None] + flow[:, 0]) / ((W - 1) / 2.0) - 1
    gy = (ys[None] + flow[:, 1]) / ((H - 1) / 2.0) - 1
    grid = torch.from_numpy(np.stack([gx, gy], -1).astype(np.float32))
    ref = torch.nn.functional.grid_sample(torch.from_numpy(feat), grid, mode="bilinear", padding_mode="zeros",
                                          align_corners=True).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=2e-4)


def test_warp_large_flow_leaves_map(hip):
    rs = np.random.RandomState(5)
    feat = rs.randn(1, 8, 38, 63).astype(np.float32)
    flow = rs.uniform(-80, 80, (1, 2, 38, 63)).astype(np.float32)
    flow[0, :, 0, :5] = 1e9
    flow[0, :, 1, :5] = -1e9
    want = oracle.warp_bilinear(feat, flow)
    got = hip.warp_bilinear(t(feat), t(flow)).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    assert (got[0, :, 0, :5] == 0).all()


def test_warp_key_path_scale_map(hip):
    rs = np.random.RandomState(6)
    feat = rs.randn(1, 1024, 38, 63).astype(np.float32)
    flow = smooth_flow(rs, 1, 38, 63, 1.5)
    mul = (1 + 0.1 * rs.randn(1, 1024, 38, 63)).astype(np.float32)
    want = oracle.warp_bilinear(feat, flow, mul=mul)
    got = hip.warp_bilinear(t(feat), t(flow), mul=t(mul)).cpu().numpy()
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("hw", [(38, 63), (36, 63), (9, 7)])
def test_warp_cur_path_fused_epilogue(hip, hw):
    H, W = hw
    C = 1024 if H > 20 else 24
    rs = np.random.RandomState(7 + H)
    feat = rs.randn(1, C, H, W).astype(np.float32)
    flow = smooth_flow(rs, 1, H, W, 2.5)
    res = (4 * rs.randn(1, 3, H, W)).astype(np.float32)
    res_w = (0.01 * rs.randn(C, 3, 1, 1)).astype(np.float32)
    res_b = (0.01 * rs.randn(C)).astype(np.float32)
    add = rs.randn(1, C, H, W).astype(np.float32)
    want = oracle.warp_bilinear(feat, flow, add=add, res=res, res_w=res_w, res_b=res_b)
    got = hip.warp_bilinear(t(feat), t(flow), add=t(add), res=t(res), res_w=t(res_w), res_b=t(res_b)).cpu().numpy()
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("shape", [(1, 1, 1024, 38, 63), (6, 2, 1024, 38, 63), (3, 1, 24, 12, 20), (2, 2, 512, 9, 7)])
def test_warp_channels_last_equals_the_operator_layout(hip, shape):
    """r6 (lsfa_warp_bilinear_cl): the non-key path's warp on channels-last maps - key feature (feat_n, H, W, C), small-net feature and output
    (N, H, W, C) - gives the ORACLE's bits, transposed; with and without the residual / add operands, feat_n dividing N (lock-step clips),
    flows that leave the map (border taps contribute 0), and max|out| in the amax slots."""
    N, feat_n, C, H, W = shape
    rs = np.random.RandomState(N * 7 + C)
    feat = rs.randn(feat_n, C, H, W).astype(np.float32)
    flow = smooth_flow(rs, N, H, W, 3.5)
    flow[:, :, 0, :] -= 4.0                                          # the first row samples above the map
    res = (4 * rs.randn(N, 3, H, W)).astype(np.float32)
    res_w = (0.01 * rs.randn(C, 3, 1, 1)).astype(np.float32)
    res_b = (0.01 * rs.randn(C)).astype(np.float32)
    add = rs.randn(N, C, H, W).astype(np.float32)
    cl = lambda a: t(np.ascontiguousarray(a.transpose(0, 2, 3, 1)))
    feats = np.concatenate([feat] * (N // feat_n), 0) if feat_n > 1 else feat      # map n samples feature n mod feat_n
    want = oracle.warp_bilinear(feats, flow, add=add, res=res, res_w=res_w, res_b=res_b)
    slots = hip.amax_slots(1, DEV)[0]
    got = hip.warp_bilinear_cl(cl(feat), t(flow), add_cl=cl(add), res=t(res), res_w=t(res_w), res_b=t(res_b), amax_out=slots)
    np.testing.assert_array_equal(got.permute(0, 3, 1, 2).cpu().numpy(), want)
    assert slots.view(torch.float32).max().item() == float(np.abs(want).max())
    if C >= 8:                                                     # the maximum over the upper half of the channels only (what the R-FCN convolution reads)
        slots2 = hip.amax_slots(1, DEV)[0]
        hip.warp_bilinear_cl(cl(feat), t(flow), add_cl=cl(add), res=t(res), res_w=t(res_w), res_b=t(res_b), amax_out=slots2, amax_c0=C // 2)
        assert slots2.view(torch.float32).max().item() == float(np.abs(want[:, C // 2:]).max())
    want_plain = oracle.warp_bilinear(feats, flow)
    np.testing.assert_array_equal(hip.warp_bilinear_cl(cl(feat), t(flow)).permute(0, 3, 1, 2).cpu().numpy(), want_plain)
    want_add = oracle.warp_bilinear(feats, flow, add=add)
    np.testing.assert_array_equal(hip.warp_bilinear_cl(cl(feat), t(flow), add_cl=cl(add)).permute(0, 3, 1, 2).cpu().numpy(), want_add)


def test_warp_broadcast_key_feature_batch(hip):
    rs = np.random.RandomState(8)
    feat = rs.randn(1, 32, 12, 20).astype(np.float32)
    flow = smooth_flow(rs, 3, 12, 20, 2.0)
    mul = rs.rand(3, 32, 12, 20).astype(np.float32)
    want = oracle.warp_bilinear(feat, flow, mul=mul)
    got = hip.warp_bilinear(t(feat), t(flow), mul=t(mul)).cpu().numpy()
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("hw", [(12, 20), (38, 63)])
def test_warp_feature_batch_that_divides_the_map_batch(hip, hw):
    """feat_n = 2 features, N = 6 maps (three frames of two lock-step clips, frame-major): map n samples feature n mod 2 - equal bit for
    bit to the oracle's warp of each map with its own feature, through the gather kernel (12 x 20) and the LDS-staged one (38 x 63), with
    the non-key epilogue (add + residual)."""
    H, W = hw
    rs = np.random.RandomState(H)
    C, B, N = 24, 2, 6
    feat = rs.randn(B, C, H, W).astype(np.float32)
    flow = smooth_flow(rs, N, H, W, 2.0)
    add = rs.randn(N, C, H, W).astype(np.float32)
    res = rs.randn(N, 3, H, W).astype(np.float32)
    rw, rb = (0.01 * rs.randn(C, 3, 1, 1)).astype(np.float32), rs.randn(C).astype(np.float32)
    got = hip.warp_bilinear(t(feat), t(flow), add=t(add), res=t(res), res_w=t(rw), res_b=t(rb)).cpu().numpy()
    for n in range(N):
        want = oracle.warp_bilinear(feat[n % B:n % B + 1], flow[n:n + 1], add=add[n:n + 1], res=res[n:n + 1], res_w=rw, res_b=rb)
        np.testing.assert_array_equal(got[n:n + 1], want)
    with pytest.raises(hip.LsfaError):
        hip.warp_bilinear(t(feat), t(flow[:5]))


@pytest.mark.parametrize("case", ["plain", "key", "cur", "mul+add+res4", "border", "broadcast", "in-place", "45x80", "26x40"])
def test_warp_staged_kernel_matches_the_oracle_and_the_gather_kernel(hip, case):
    """r3: planes staged in LDS by DMA (warp_staged_kernel).  Forced ('staged' makes an unsupported shape an error), on every
    operand combination, with flows that leave the map on all sides, odd plane alignments (38 x 63 planes start 8 bytes off every
    other channel), the broadcast key feature, the in-place `add`, a map larger than one 640-thread pass and one smaller than the
    default threshold: bit-identical to the oracle and to the gather kernel."""
    H, W = {"45x80": (45, 80), "26x40": (26, 40)}.get(case, (38, 63))
    N, C = (3, 40) if case in ("broadcast", "45x80", "26x40") else (2, 24)
    rs = np.random.RandomState(len(case) + H)
    feat = rs.randn(1 if case == "broadcast" else N, C, H, W).astype(np.float32)
    flow = smooth_flow(rs, N, H, W, 2.5)
    if case == "border":
        flow = rs.uniform(-70, 70, (N, 2, H, W)).astype(np.float32)
        flow[0, :, 0, :7] = 1e9
        flow[1, :, -1, -7:] = -1e9
        flow[0, 0, 5, :] = -1.5          # x0 = -2 .. -1: the left column half in
        flow[0, 1, :, 9] = 1.25
    kw = {}
    if case in ("key", "mul+add+res4", "broadcast"):
        kw["mul"] = (1 + 0.1 * rs.randn(N, C, H, W)).astype(np.float32)
    if case in ("cur", "mul+add+res4", "border", "in-place", "45x80", "26x40"):
        kw["add"] = rs.randn(N, C, H, W).astype(np.float32)
    if case in ("cur", "mul+add+res4", "border", "45x80"):
        rc = 4 if case == "mul+add+res4" else 3
        kw["res"] = (4 * rs.randn(N, rc, H, W)).astype(np.float32)
        kw["res_w"] = (0.01 * rs.randn(C, rc, 1, 1)).astype(np.float32)
        kw["res_b"] = (0.01 * rs.randn(C)).astype(np.float32)
    want = oracle.warp_bilinear(feat, flow, **kw)
    dkw = {k: t(v) for k, v in kw.items()}
    got = {}
    try:
        for variant in ("staged", "gather"):
            hip.warp_set_variant(variant)
            if case == "in-place":
                buf = dkw["add"].clone()
                got[variant] = hip.warp_bilinear(t(feat), t(flow), add=buf, out=buf).cpu().numpy()
            else:
                got[variant] = hip.warp_bilinear(t(feat), t(flow), **dkw).cpu().numpy()
    finally:
        hip.warp_set_variant("auto")
    np.testing.assert_array_equal(got["staged"], want)
    np.testing.assert_array_equal(got["gather"], want)


def test_warp_staged_kernel_refuses_what_it_cannot_take(hip):
    feat, flow = torch.randn(1, 8, 9, 7, device=DEV), torch.zeros(1, 2, 9, 7, device=DEV)      # 63 pixels: odd
    hip.warp_set_variant("staged")
    try:
        with pytest.raises(hip.LsfaError):
            hip.warp_bilinear(feat, flow)
    finally:
        hip.warp_set_variant("auto")
    assert hip.warp_bilinear(feat, flow).shape == feat.shape


def test_warp_32_maps_per_launch_bit_exact(hip):
    """BASELINE configs[4]'s launch shape (32 maps x 1024 channels): the many-planes instance of the staged kernel against the
    gather kernel (the oracle at this size takes minutes; both kernels are checked against it at small sizes above)."""
    g = torch.Generator(device=DEV).manual_seed(3)
    feat = torch.randn((32, 1024, 38, 63), device=DEV, generator=g)
    mul = torch.randn((32, 1024, 38, 63), device=DEV, generator=g)
    flow = torch.randn((32, 2, 38, 63), device=DEV, generator=g) * 3
    try:
        hip.warp_set_variant("staged")
        a = hip.warp_bilinear(feat, flow, mul=mul)
        hip.warp_set_variant("gather")
        b = hip.warp_bilinear(feat, flow, mul=mul)
    finally:
        hip.warp_set_variant("auto")
    assert torch.equal(a, b)


def test_warp_identity_property(hip):
    feat = torch.randn(1, 1024, 38, 63, device=DEV)
    out = hip.warp_bilinear(feat, torch.zeros(1, 2, 38, 63, device=DEV))
    # zero flow samples integer positions (up to the normalise/denormalise round trip)
    assert torch.allclose(out, feat, rtol=0, atol=2e-5 * feat.abs().max().item())


# ------------------------------------------------------------------ aggregate ---------
@pytest.mark.parametrize("shape", [(1024, 38, 63), (1024, 36, 63), (20, 8, 8), (3, 5, 7)])
def test_aggregate_softmax2_bit_exact(hip, shape):
    C, H, W = shape
    rs = np.random.RandomState(C + W)
    a, b = rs.randn(1, C, H, W).astype(np.float32), rs.randn(1, C, H, W).astype(np.float32)
    logits = (3 * rs.randn(2, 1, H, W)).astype(np.float32)
    logits[0, 0, 0, 0], logits[1, 0, 0, 0] = 100.0, -100.0
    want = oracle.aggregate_softmax2(a, b, logits)
    got = hip.aggregate_softmax2(t(a), t(b), t(logits)).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    ref = torch.softmax(torch.from_numpy(logits), 0)
    np.testing.assert_allclose(got, (ref[0] * torch.from_numpy(a) + ref[1] * torch.from_numpy(b)).numpy(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", [(1, 1024, 38, 63), (3, 20, 8, 8)])
def test_aggregate_softmax2_reads_logit_rows_in_place(hip, shape):
    """r5 (lsfa_aggregate_softmax2_rows): the 2N rows of logits are read where the Nq net's last convolution left them - rows of a wider
    buffer, `logit_row_stride` floats apart - instead of from a gathered (2N, 1, H, W) copy: the same bits as the packed operator."""
    N, C, H, W = shape
    rs = np.random.RandomState(7 * N + C)
    a, b = rs.randn(N, C, H, W).astype(np.float32), rs.randn(N, C, H, W).astype(np.float32)
    logits = (3 * rs.randn(2 * N, 1, H, W)).astype(np.float32)
    want = oracle.aggregate_softmax2(a, b, logits)
    stride = H * W + 37                                            # rows inside a wider buffer, garbage in between
    wide = np.full((2 * N, stride), np.nan, np.float32)
    wide[:, :H * W] = logits.reshape(2 * N, H * W)
    got = hip.aggregate_softmax2(t(a), t(b), t(wide), logit_row_stride=stride).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    with pytest.raises(hip.LsfaError):                             # a buffer that does not hold the last row
        hip.aggregate_softmax2(t(a), t(b), t(wide.reshape(-1)[:(2 * N - 1) * stride + H * W - 1].copy()), logit_row_stride=stride)
    with pytest.raises(hip.LsfaError):                             # rows that overlap
        hip.aggregate_softmax2(t(a), t(b), t(wide), logit_row_stride=H * W - 1)
    with pytest.raises(hip.LsfaError):                             # logits on another device than the maps: the C side cannot tell
        hip.aggregate_softmax2(t(a), t(b), torch.from_numpy(wide), logit_row_stride=stride)


@pytest.mark.parametrize("shape", [(4, 1024, 38, 63), (3, 20, 8, 8), (32, 64, 38, 63)])
def test_aggregate_softmax2_batched_bit_exact(hip, shape):
    """N maps per launch (several clips advancing together; the HBM-resident roofline mode): map n uses logits
    rows n and N + n, each map bit-identical to the N = 1 operator."""
    N, C, H, W = shape
    rs = np.random.RandomState(N + C)
    a, b = rs.randn(N, C, H, W).astype(np.float32), rs.randn(N, C, H, W).astype(np.float32)
    logits = (3 * rs.randn(2 * N, 1, H, W)).astype(np.float32)
    got = hip.aggregate_softmax2(t(a), t(b), t(logits)).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.aggregate_softmax2(a, b, logits))
    one = hip.aggregate_softmax2(t(a[1:2]), t(b[1:2]), t(np.stack([logits[1], logits[N + 1]]))).cpu().numpy()
    np.testing.assert_array_equal(got[1:2], one)


@pytest.mark.parametrize("shape", [(1024, 2048, 38, 63), (64, 128, 12, 14), (16, 100, 5, 7), (8, 37, 3, 3)])
def test_aggregate_cosine_bit_exact(hip, shape):
    """Fgfa weights at LSFA's real shape (2048-channel embeddings on the 38x63 map) and at shapes with E not a
    multiple of the 64 lanes / pixel counts not a multiple of the workgroup tile: the lane-split reduction
    follows the oracle's fixed tree, so the output is equal bit for bit; a float64 statement of the same
    formula bounds what the tree order can differ by."""
    C, E, H, W = shape
    rs = np.random.RandomState(9 + E)
    a, b = rs.randn(1, C, H, W).astype(np.float32), rs.randn(1, C, H, W).astype(np.float32)
    ew, ec = rs.randn(1, E, H, W).astype(np.float32), rs.randn(1, E, H, W).astype(np.float32)
    ew[0, :, 0, 0] = ec[0, :, 0, 0]                 # identical embeddings at one pixel: both logits 1 -> weights 1/2
    want = oracle.aggregate_cosine(a, b, ew, ec)
    got = hip.aggregate_cosine(t(a), t(b), t(ew), t(ec)).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    e64w, e64c = ew[0].astype(np.float64), ec[0].astype(np.float64)
    nw, nc = np.sqrt((e64w ** 2).sum(0) + 1e-10), np.sqrt((e64c ** 2).sum(0) + 1e-10)
    l0, l1 = ((e64w / nw) * (e64c / nc)).sum(0), ((e64c / nc) ** 2).sum(0)
    w0 = 1.0 / (1.0 + np.exp(l1 - l0))
    np.testing.assert_allclose(got[0], w0 * a[0] + (1 - w0) * b[0], rtol=0, atol=2e-5)


# ------------------------------------------------------------------ NMS ---------------
def clustered_dets(rs, n, dtype=np.float32):
    k = max(3, n // 25)
    cx = rs.uniform(50, 950, k)[rs.randint(0, k, n)] + rs.normal(0, 12, n)
    cy = rs.uniform(50, 550, k)[rs.randint(0, k, n)] + rs.normal(0, 12, n)
    w, h = rs.uniform(20, 220, n), rs.uniform(20, 220, n)
    x1, y1 = np.clip(cx - w / 2, 0, 999), np.clip(cy - h / 2, 0, 599)
    x2, y2 = np.clip(cx + w / 2, 0, 999), np.clip(cy + h / 2, 0, 599)
    s = np.sort(rs.rand(n))[::-1]
    return np.stack([x1, y1, x2, y2, s], 1).astype(dtype)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 300, 1000, 6000])
@pytest.mark.parametrize("th", [0.3, 0.7])
def test_nms_sorted_bit_exact(hip, n, th):
    dets = clustered_dets(np.random.RandomState(n), n)
    want = oracle.nms_sorted(dets, th)
    keep, num = hip.nms_sorted(t(dets), th)
    k = int(num.item())
    assert k == len(want)
    np.testing.assert_array_equal(keep[:k].cpu().numpy(), want)


def test_nms_golden_reference_survivors(hip, golden):
    """The HIP NMS reproduces the REFERENCE's numpy nms survivors on the golden boxes."""
    for n in (50, 300, 2000):
        for th in (0.3, 0.7):
            dets = golden["g2_n%d_float32_dets" % n]
            want = golden["g2_n%d_float32_keep_%02d" % (n, int(th * 10))]
            order = np.argsort(-dets[:, 4], kind="stable")
            keep, num = hip.nms_sorted(t(dets[order]), th)
            got = order[keep[:int(num.item())].cpu().numpy()]
            np.testing.assert_array_equal(got, want)
            # and through the reference's own host-pointer C entry point `_nms`
            np.testing.assert_array_equal(order[hip.nms_host(dets[order], th)], want)


def test_nms_wrappers_reference_interface(hip, golden):
    """lib/nms/nms.py's wrappers (py_ / cpu_ / gpu_nms_wrapper, gpu_nms) over `_nms`: the reference's
    survivor lists for its own float32 AND float64 golden boxes (IoUs computed in float32 here)."""
    from lsfa_amd.nms.nms import py_nms_wrapper, cpu_nms_wrapper, gpu_nms_wrapper, gpu_nms
    for n in (50, 300, 2000):
        for th in (0.3, 0.7):
            for dt in ("float32", "float64"):
                dets = golden["g2_n%d_%s_dets" % (n, dt)]
                want = golden["g2_n%d_%s_keep_%02d" % (n, dt, int(th * 10))]
                for f in (py_nms_wrapper(th), cpu_nms_wrapper(th), gpu_nms_wrapper(th, 0)):
                    np.testing.assert_array_equal(np.asarray(f(dets)), want)
    assert gpu_nms(np.zeros((0, 5), np.float32), 0.3) == []


def test_py_nms_wrapper_computes_in_the_dets_dtype(hip):
    """lib/nms/nms.py:37-74 computes IoU in the dtype of `dets` — float64 in pred_eval.  Two float64 boxes whose
    IoU sits 1e-9 below / above the threshold: float32 arithmetic cannot tell the two cases apart, the float64
    kernel (lsfa_nms_sorted_f64) must, and agree with the oracle's float64 statement of the numpy loop."""
    from lsfa_amd.nms.nms import py_nms_wrapper, cpu_nms_wrapper, gpu_nms_wrapper
    s0 = 100.0 - 6000.0 / 1.3 / 100.0            # horizontal shift at which IoU of two 100x100 boxes is exactly 0.3
    outcomes = []
    for eps in (+1e-7, -1e-7):
        dets = np.array([[0, 0, 99, 99, 0.9], [s0 + eps, 0, 99 + s0 + eps, 99, 0.8]], np.float64)
        want = list(oracle.nms_f64(dets, 0.3))
        assert list(py_nms_wrapper(0.3)(dets)) == want and list(cpu_nms_wrapper(0.3)(dets)) == want
        outcomes.append((len(want), len(gpu_nms_wrapper(0.3, 0)(dets))))
    assert outcomes[0][0] == 2 and outcomes[1][0] == 1          # float64: kept / suppressed
    assert outcomes[0][1] == outcomes[1][1]                     # float32 (_nms): the same answer for both
    # random float64 clusters, ties excluded by construction
    rs = np.random.RandomState(21)
    for n in (1, 63, 64, 65, 700):
        dets = clustered_dets(rs, n, np.float64)
        dets[:, :4] += rs.uniform(0, 1e-3, (n, 4))              # off the float32 grid
        np.testing.assert_array_equal(np.asarray(py_nms_wrapper(0.45)(dets)), oracle.nms_f64(dets, 0.45))


def test_nms_all_identical_boxes_and_empty(hip):
    dets = np.tile(np.array([[10, 10, 50, 50, 0.5]], np.float32), (200, 1))
    keep, num = hip.nms_sorted(t(dets), 0.5)
    assert int(num.item()) == 1 and int(keep[0].item()) == 0
    keep, num = hip.nms_sorted(torch.zeros(0, 5, device=DEV), 0.5)
    assert int(num.item()) == 0


# ------------------------------------------------------------------ Proposal ----------
def rpn_inputs(rs, B, H, W, A=9, frac_pos=0.02):
    logit = rs.randn(B, 2, A * H, W).astype(np.float32) * 2
    logit[:, 1] += np.log(frac_pos / (1 - frac_pos))
    e = np.exp(logit - logit.max(1, keepdims=True))
    prob = (e / e.sum(1, keepdims=True)).astype(np.float32).reshape(B, 2 * A, H, W)
    deltas = (rs.randn(B, 4 * A, H, W) * np.tile([0.1, 0.1, 0.4, 0.4], A)[None, :, None, None]).astype(np.float32)
    return prob, deltas


@pytest.mark.parametrize("cfg", [
    dict(H=38, W=63, im=(600, 1000, 1.0), pre=6000, post=300, min_size=0),
    dict(H=36, W=63, im=(562, 1000, 0.78125), pre=6000, post=300, min_size=0),
    dict(H=38, W=63, im=(600, 1000, 1.6), pre=6000, post=300, min_size=16),
    dict(H=10, W=12, im=(160, 192, 1.0), pre=6000, post=300, min_size=0),   # pre_n clamps to 1080
    dict(H=38, W=63, im=(600, 1000, 1.0), pre=12000, post=1000, min_size=0),
])
def test_proposal_bit_exact(hip, cfg):
    H, W = cfg["H"], cfg["W"]
    rs = np.random.RandomState(H * W + cfg["pre"])
    prob, deltas = rpn_inputs(rs, 1, H, W)
    im_info = np.array([cfg["im"]], np.float32)
    want_rois, want_scores = oracle.proposal(prob, deltas, im_info, rpn_pre_nms_top_n=cfg["pre"],
                                             rpn_post_nms_top_n=cfg["post"], rpn_min_size=cfg["min_size"])
    op = hip.ProposalOp(rpn_pre_nms_top_n=cfg["pre"], rpn_post_nms_top_n=cfg["post"], threshold=0.7,
                        rpn_min_size=cfg["min_size"], output_score=True)
    rois, scores = op(t(prob), t(deltas), t(im_info))
    np.testing.assert_array_equal(rois.cpu().numpy(), want_rois)
    np.testing.assert_array_equal(scores.cpu().numpy(), want_scores)


def test_proposal_ties_stable_order_and_cyclic_pad(hip):
    """Heavy score ties (quantised scores + the -1 block outside the real image) exercise the
    (score desc, anchor index asc) order; few survivors exercise the cyclic pad (multi_proposal.cu:381)."""
    rs = np.random.RandomState(4)
    H, W = 38, 63
    prob, deltas = rpn_inputs(rs, 1, H, W)
    prob = (np.round(prob * 20) / 20).astype(np.float32)       # 21 distinct score values
    deltas[:] = 0                                               # identical anchors per cell -> massive overlap
    im_info = np.array([[100, 160, 1.0]], np.float32)           # most cells outside the real image -> -1
    want_rois, want_scores, order, keep, nkeep = oracle.proposal(prob, deltas, im_info, return_debug=True)
    op = hip.ProposalOp(rpn_min_size=0, output_score=True)
    rois, scores = op(t(prob), t(deltas), t(im_info))
    np.testing.assert_array_equal(rois.cpu().numpy(), want_rois)
    np.testing.assert_array_equal(scores.cpu().numpy(), want_scores)
    assert nkeep[0] < 300                                       # the pad path really ran


@pytest.mark.parametrize("kind", ["flat", "all_equal", "logits", "few_valid", "saturated", "huge_boxes"])
def test_proposal_score_distributions_that_stress_the_select(hip, kind):
    """The chip-wide plan picks its candidates by a 4096-bin histogram of the order keys and ranks them
    exactly; these inputs put thousands of keys into the threshold bin (nearly flat scores, all scores equal,
    the -1 block larger than what is left), or outside the range the bins resolve (raw logits, scores of
    exactly 1.0).  Results must still equal the oracle's stable sort + NMS bit for bit."""
    rs = np.random.RandomState({"flat": 1, "all_equal": 2, "logits": 3, "few_valid": 4, "saturated": 5, "huge_boxes": 6}[kind])
    H, W = 38, 63
    prob, deltas = rpn_inputs(rs, 1, H, W)
    im_info = np.array([[600, 1000, 1.0]], np.float32)
    min_size = 0
    if kind == "flat":
        prob = (0.02 + 0.002 * rs.rand(*prob.shape)).astype(np.float32)
    elif kind == "all_equal":
        prob[:] = 0.25
    elif kind == "logits":
        prob = (rs.randn(*prob.shape) * 6).astype(np.float32)           # negative and > 1 "scores"
    elif kind == "few_valid":
        im_info = np.array([[600, 1000, 4.0]], np.float32)                # min_size 16 * 4 = 64 px: most anchors get -1
        min_size = 16
    elif kind == "saturated":
        prob[:, 9:] = np.where(rs.rand(1, 9, H, W) < 0.4, 1.0, prob[:, 9:]).astype(np.float32)
    elif kind == "huge_boxes":       # every box clips to nearly the whole image: each 64-box block is one pile of
        deltas[:, 2::4] = 3.0         # near-duplicates, the sweep visits all 94 blocks and keeps a handful
        deltas[:, 3::4] = 3.0
    want_rois, want_scores = oracle.proposal(prob, deltas, im_info, rpn_min_size=min_size)
    op = hip.ProposalOp(rpn_min_size=min_size, output_score=True)
    rois, scores = op(t(prob), t(deltas), t(im_info))
    np.testing.assert_array_equal(rois.cpu().numpy(), want_rois)
    np.testing.assert_array_equal(scores.cpu().numpy(), want_scores)


@pytest.mark.parametrize("plan", ["single", "chip", "chip-box-sweep"])
def test_proposal_launch_plans_agree(hip, plan):
    """lsfa_proposal_set_plan: the single-workgroup plan and the chip-wide plan give the oracle's result bit for bit
    (the default picks by shape; test_proposal_bit_exact's pre_n = 12000 case runs the single-workgroup plan anyway)."""
    rs = np.random.RandomState(77)
    prob, deltas = rpn_inputs(rs, 2, 38, 63)
    im_info = np.array([[600, 1000, 1.0], [592, 990, 1.3]], np.float32)
    want_rois, want_scores = oracle.proposal(prob, deltas, im_info, rpn_min_size=16)
    hip.proposal_set_plan(plan)
    try:
        rois, scores = hip.ProposalOp(rpn_min_size=16, output_score=True)(t(prob), t(deltas), t(im_info))
    finally:
        hip.proposal_set_plan('auto')
    np.testing.assert_array_equal(rois.cpu().numpy(), want_rois)
    np.testing.assert_array_equal(scores.cpu().numpy(), want_scores)


def test_proposal_multi_image(hip):
    rs = np.random.RandomState(12)
    prob, deltas = rpn_inputs(rs, 3, 20, 30)
    im_info = np.array([[320, 480, 1.0], [300, 470, 0.9], [320, 480, 1.2]], np.float32)
    want_rois, want_scores = oracle.proposal(prob, deltas, im_info, rpn_pre_nms_top_n=3000, rpn_post_nms_top_n=200)
    op = hip.ProposalOp(rpn_pre_nms_top_n=3000, rpn_post_nms_top_n=200, rpn_min_size=0, output_score=True)
    rois, scores = op(t(prob), t(deltas), t(im_info))
    np.testing.assert_array_equal(rois.cpu().numpy(), want_rois)
    np.testing.assert_array_equal(scores.cpu().numpy(), want_scores)


def test_proposal_errors(hip):
    op = hip.ProposalOp(scales=(8, 16), ratios=(0.5, 1, 2))
    with pytest.raises(hip.LsfaError):  # CHECK_EQ(num_anchors, ratios*scales), multi_proposal.cu:446
        op(torch.zeros(1, 18, 4, 4, device=DEV), torch.zeros(1, 36, 4, 4, device=DEV), torch.ones(1, 3, device=DEV))


# ------------------------------------------------------------------ det post ----------
def test_det_postprocess_golden(hip, golden):
    """HIP frame post-processing == the reference's bbox_pred/clip_boxes/nms loop (golden G5)."""
    rois, deltas, probs = golden["g5_rois"], golden["g5_deltas"], golden["g5_probs"]
    scale = float(golden["g5_scale"])
    dets, counts, keep_idx = hip.det_postprocess(t(rois), t(deltas), t(probs), 600, 1000, scale)
    counts = counts.cpu().numpy()
    np.testing.assert_array_equal(counts, golden["g5_counts"])
    got = np.vstack([dets[j, :counts[j]].cpu().numpy() for j in range(31)])
    np.testing.assert_array_equal(got[:, 4], golden["g5_dets"][:, 4])
    np.testing.assert_allclose(got[:, :4], golden["g5_dets"][:, :4], rtol=2e-6, atol=2e-4)
    pred = hip.bbox_pred_clip(t(rois), t(deltas), 600, 1000, scale).cpu().numpy()
    np.testing.assert_allclose(pred, golden["g5_pred_boxes"], rtol=2e-6, atol=2e-4)


@pytest.mark.parametrize("seed,cap", [(0, 300), (1, 0), (2, 50), (3, 100000)])
def test_det_postprocess_vs_oracle(hip, seed, cap):
    rs = np.random.RandomState(seed)
    R, ncls = 300, 31
    rois = rand_rois(rs, R, small=0.0)
    rois[:, 1:] = clustered_dets(rs, R)[:, :4]
    deltas = (0.15 * rs.randn(R, 8)).astype(np.float32)
    logits = (2 * rs.randn(R, ncls)).astype(np.float32)
    e = np.exp(logits - logits.max(1, keepdims=True))
    probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
    w_d, w_c, w_k = oracle.det_postprocess(rois, deltas, probs, 600, 1000, 1.3, max_per_image=cap)
    dets, counts, keep_idx = hip.det_postprocess(t(rois), t(deltas), t(probs), 600, 1000, 1.3, max_per_image=cap)
    counts = counts.cpu().numpy()
    np.testing.assert_array_equal(counts, w_c)
    for j in range(ncls):
        np.testing.assert_array_equal(keep_idx[j, :counts[j]].cpu().numpy(), w_k[j, :counts[j]])
        # fp64 arithmetic in the same order on both sides; exp() of the two libms may differ by an ulp
        np.testing.assert_allclose(dets[j, :counts[j]].cpu().numpy(), w_d[j, :counts[j]], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("pieces", [3, 1])
@pytest.mark.parametrize("shape", [(1, 1024, 38, 63), (3, 512, 12, 20), (2, 640, 7, 9)])
def test_rpn_head_on_the_nchw_map(hip, shape, pieces):
    """The RPN head: both 1x1 convolutions as one lsfa_conv_fwd on channels [0, 512) of an NCHW map (x_nchw: the direct kernel's K-major
    operand form; three exact bf16 pieces = fp32 accuracy, one piece = the bf16 mode) + lsfa_rpn_softmax_split, against float64; the
    probabilities of an anchor's two classes add up to one; maps whose pixel count is not a multiple of the tiles; channels past 512 are
    not read; convolutions the K-major form does not take are refused."""
    N, C, H, W = shape
    A = 9
    g = torch.Generator(device=DEV).manual_seed(C + H)
    feat = torch.randn((N, C, H, W), device=DEV, generator=g) * 2.0
    w = torch.randn((6 * A, 512), device=DEV, generator=g) * 0.05
    b = torch.randn(6 * A, device=DEV, generator=g) * 0.1
    w64 = torch.zeros((64, 512, 1, 1), device=DEV)
    w64[:6 * A, :, 0, 0] = w
    b64 = torch.zeros(64, device=DEV)
    b64[:6 * A] = b
    sw = hip.SplitWeight(w64, real_cout=6 * A, pieces=pieces)
    logits_g = hip.conv_split(feat, sw, b64, x_nchw=True)
    assert logits_g.shape == (N, H, W, 64)
    cls_prob, bbox = hip.rpn_softmax_split(logits_g, A)
    x = feat[:, :512].double().cpu()
    logits = torch.einsum('oc,nchw->nohw', w.double().cpu(), x) + b.double().cpu().view(1, -1, 1, 1)
    want_p = torch.softmax(logits[:, :2 * A].reshape(N, 2, A * H, W), dim=1).reshape(N, 2 * A, H, W)
    tol = (2e-2 if pieces == 1 else 2e-6 * 512 ** 0.5) * float(logits.abs().max())
    assert float((bbox.double().cpu() - logits[:, 2 * A:]).abs().max()) < tol
    assert float((cls_prob.double().cpu() - want_p).abs().max()) < tol
    assert float((cls_prob[:, :A] + cls_prob[:, A:] - 1.0).abs().max()) < 3e-7
    # the same numbers as the channels-last form of the same convolution
    rows = hip.conv_split(feat[:, :512].permute(0, 2, 3, 1).contiguous(), sw, b64)
    assert torch.equal(rows, logits_g)
    if C > 512:
        feat2 = feat.clone()
        feat2[:, 512:] = float('nan')
        assert torch.equal(hip.conv_split(feat2, sw, b64, x_nchw=True), logits_g)
    with pytest.raises(hip.LsfaError):
        hip.rpn_softmax_split(logits_g, 11)
    with pytest.raises(hip.LsfaError):
        hip.conv_split(feat, hip.SplitWeight(torch.zeros((64, 512, 3, 3), device=DEV), pieces=pieces), None, 1, 1, 1, x_nchw=True)


@pytest.mark.parametrize("cap", [300, 40])
def test_det_postprocess_batch_equals_the_oracle_per_image(hip, cap):
    """lsfa_det_postprocess_batch (the frames of a segment / the clips of a lock-step batch in one launch pair): image b of five, each
    with its own rois / deltas / probs (one with nothing above threshold), equals the oracle run on that image alone and the single-image
    entry point bit for bit."""
    rs = np.random.RandomState(40 + cap)
    B, R, ncls = 5, 300, 31
    rois = np.concatenate([rand_rois(rs, R, small=0.0) for _ in range(B)], 0)
    for b in range(B):
        rois[b * R:(b + 1) * R, 0] = b
        rois[b * R:(b + 1) * R, 1:] = clustered_dets(rs, R)[:, :4]
    deltas = (0.15 * rs.randn(B * R, 8)).astype(np.float32)
    logits = (2 * rs.randn(B * R, ncls)).astype(np.float32)
    e = np.exp(logits - logits.max(1, keepdims=True))
    probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
    probs[3 * R:4 * R, 1:] = np.float32(1e-6)          # image 3: every class below the threshold
    out = (torch.full((B, ncls, R, 5), -7.0, dtype=torch.float64, device=DEV), torch.full((B, ncls), -7, dtype=torch.int32, device=DEV),
           torch.full((B, ncls, R), -7, dtype=torch.int32, device=DEV))
    dets, counts, keep_idx = hip.det_postprocess_batch(t(rois), t(deltas), t(probs), B, 600, 1000, 1.3, out, max_per_image=cap)
    assert int(counts[3].sum()) == 0
    for b in range(B):
        sl = slice(b * R, (b + 1) * R)
        w_d, w_c, w_k = oracle.det_postprocess(rois[sl], deltas[sl], probs[sl], 600, 1000, 1.3, max_per_image=cap)
        s_d, s_c, s_k = hip.det_postprocess(t(rois[sl]), t(deltas[sl]), t(probs[sl]), 600, 1000, 1.3, max_per_image=cap)
        c = counts[b].cpu().numpy()
        np.testing.assert_array_equal(c, w_c)
        assert torch.equal(counts[b], s_c)
        for j in range(ncls):
            np.testing.assert_array_equal(keep_idx[b, j, :c[j]].cpu().numpy(), w_k[j, :c[j]])
            assert torch.equal(keep_idx[b, j, :c[j]], s_k[j, :c[j]]) and torch.equal(dets[b, j, :c[j]], s_d[j, :c[j]])
            np.testing.assert_allclose(dets[b, j, :c[j]].cpu().numpy(), w_d[j, :c[j]], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("kind", ["piles", "equal_scores", "tied_at_cap", "two_values", "disjoint"])
def test_det_postprocess_inputs_that_stress_sweep_and_cap(hip, kind):
    """Near-duplicate piles (suppression chains as long as a 64-box block: the sweep's scalar scan), thousands of
    equal scores (the cap's threshold bin overflows its list: radix fallback), ties exactly at the cap threshold
    (all of them are kept, tester.py:277-281), no overlaps at all (every box survives, the fixpoint settles at once)."""
    rs = np.random.RandomState(11)
    R, ncls = 300, 31
    rois = rand_rois(rs, R, small=0.0)
    deltas = (0.05 * rs.randn(R, 8)).astype(np.float32)
    logits = (2 * rs.randn(R, ncls)).astype(np.float32)
    e = np.exp(logits - logits.max(1, keepdims=True))
    probs = (e / e.sum(1, keepdims=True)).astype(np.float32)
    cap = 300
    if kind == "piles":          # 5 piles of 60 boxes, each box a slightly shifted copy of its neighbour
        for p_ in range(5):
            x, y = 100 + 150 * p_, 80 + 60 * p_
            for i in range(60):
                rois[p_ * 60 + i, 1:] = [x + 2.5 * i, y + 1.5 * i, x + 2.5 * i + 180, y + 1.5 * i + 140]
        deltas[:] = 0
        probs[:] = np.float32(1.0 / ncls)
        probs[:, 1:] += (1e-4 * rs.rand(R, ncls - 1)).astype(np.float32)
    elif kind == "equal_scores":
        probs[:] = np.float32(1.0 / ncls)
        x1 = (rs.rand(R) * 800).astype(np.float32); y1 = (rs.rand(R) * 450).astype(np.float32)
        rois[:, 1:] = np.stack([x1, y1, x1 + 30, y1 + 30], 1)
        deltas[:] = 0
    elif kind == "tied_at_cap":
        probs[:] = np.float32(1e-5)
        x1 = (np.arange(R) % 20 * 48).astype(np.float32); y1 = (np.arange(R) // 20 * 38).astype(np.float32)
        rois[:, 1:] = np.stack([x1, y1, x1 + 20, y1 + 20], 1)      # disjoint: everything above threshold survives
        deltas[:] = 0
        probs[:40, 3] = 0.9; probs[40:120, 5] = 0.5; probs[120:, 7] = 0.5       # the 50-th best sits inside a 260-way tie
        cap = 50
    elif kind == "two_values":
        probs[:] = np.float32(0.25)
        probs[::3] = np.float32(0.5)
        cap = 700
    else:
        x1 = (np.arange(R) % 20 * 48).astype(np.float32); y1 = (np.arange(R) // 20 * 38).astype(np.float32)
        rois[:, 1:] = np.stack([x1, y1, x1 + 20, y1 + 20], 1)
        deltas[:] = 0
    w_d, w_c, w_k = oracle.det_postprocess(rois, deltas, probs, 600, 1000, 1.0, max_per_image=cap)
    dets, counts, keep_idx = hip.det_postprocess(t(rois), t(deltas), t(probs), 600, 1000, 1.0, max_per_image=cap)
    counts = counts.cpu().numpy()
    np.testing.assert_array_equal(counts, w_c)
    if kind == "tied_at_cap":
        assert counts.sum() == 40 + 80 + 180
    for j in range(ncls):
        np.testing.assert_array_equal(keep_idx[j, :counts[j]].cpu().numpy(), w_k[j, :counts[j]])
        np.testing.assert_allclose(dets[j, :counts[j]].cpu().numpy(), w_d[j, :counts[j]], rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("n,th", [(640, 0.5), (2000, 0.7)])
def test_nms_near_duplicate_piles(hip, n, th):
    """Blocks that are one pile of near-duplicates: the suppression chain inside a 64-box block is ~64 long, which the
    block resolution answers by scanning survivors instead of iterating the ballot fixpoint."""
    rs = np.random.RandomState(n)
    i = np.arange(n, dtype=np.float32)
    pile = (i // 100)
    x1 = 50 + 90 * pile + 1.75 * (i % 100); y1 = 40 + 30 * pile + 1.25 * (i % 100)
    boxes = np.stack([x1, y1, x1 + 160, y1 + 120], 1).astype(np.float32)
    scores = np.sort(rs.rand(n).astype(np.float32))[::-1].copy()
    dets = np.concatenate([boxes, scores[:, None]], 1).astype(np.float32)
    want = oracle.nms_sorted(dets, th)
    keep, num = hip.nms_sorted(t(dets), th)
    k = int(num.item())
    assert k == len(want)
    np.testing.assert_array_equal(keep[:k].cpu().numpy(), want)
    d64 = dets.astype(np.float64)
    want64 = oracle.nms_f64(d64, th)
    keep, num = hip.nms_sorted_f64(t(d64), th)
    assert int(num.item()) == len(want64)
    np.testing.assert_array_equal(keep[:len(want64)].cpu().numpy(), want64)


# ------------------------------------------------------------------ DCN / BN ----------
@pytest.mark.parametrize("cfg", [dict(C=16, H=19, W=23, k=3, pad=2, stride=1, dil=2, dg=4),
                                 dict(C=8, H=20, W=21, k=3, pad=1, stride=2, dil=1, dg=4),
                                 dict(C=12, H=9, W=9, k=3, pad=1, stride=1, dil=1, dg=1)])
def test_deform_im2col_bit_exact(hip, cfg):
    rs = np.random.RandomState(cfg["C"])
    C, H, W, k = cfg["C"], cfg["H"], cfg["W"], cfg["k"]
    Ho = (H + 2 * cfg["pad"] - (cfg["dil"] * (k - 1) + 1)) // cfg["stride"] + 1
    Wo = (W + 2 * cfg["pad"] - (cfg["dil"] * (k - 1) + 1)) // cfg["stride"] + 1
    data = rs.randn(2, C, H, W).astype(np.float32)
    offset = (2.0 * rs.randn(2, 2 * k * k * cfg["dg"], Ho, Wo)).astype(np.float32)
    want = oracle.deform_im2col(data, offset, k, k, cfg["pad"], cfg["stride"], cfg["dil"], cfg["dg"])
    got = hip.deform_im2col(t(data), t(offset), k, k, cfg["pad"], cfg["stride"], cfg["dil"], cfg["dg"]).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    # zero offsets reduce to an ordinary im2col (F.unfold)
    z = hip.deform_im2col(t(data), torch.zeros_like(t(offset)), k, k, cfg["pad"], cfg["stride"], cfg["dil"], cfg["dg"])
    ref = torch.nn.functional.unfold(torch.from_numpy(data), k, dilation=cfg["dil"], padding=cfg["pad"], stride=cfg["stride"])
    np.testing.assert_array_equal(z.cpu().numpy(), ref.numpy())


@pytest.mark.parametrize("cfg", [dict(C=16, H=19, W=23, k=3, pad=2, stride=1, dil=2, dg=4),
                                 dict(C=32, H=20, W=21, k=3, pad=1, stride=2, dil=1, dg=4),
                                 dict(C=512, H=38, W=63, k=3, pad=2, stride=1, dil=2, dg=4),
                                 dict(C=12, H=9, W=9, k=3, pad=1, stride=1, dil=1, dg=1)])
def test_deform_im2col_channels_last_bit_exact(hip, cfg):
    """The channels-last kernel samples exactly what the NCHW statement samples:
    col_cl[n, pixel, tap, c] == col[n, c*KK + tap, pixel]."""
    rs = np.random.RandomState(cfg["C"] + 1)
    C, H, W, k = cfg["C"], cfg["H"], cfg["W"], cfg["k"]
    Ho = (H + 2 * cfg["pad"] - (cfg["dil"] * (k - 1) + 1)) // cfg["stride"] + 1
    Wo = (W + 2 * cfg["pad"] - (cfg["dil"] * (k - 1) + 1)) // cfg["stride"] + 1
    N = 1 if C > 100 else 2
    data = rs.randn(N, C, H, W).astype(np.float32)
    offset = (2.0 * rs.randn(N, 2 * k * k * cfg["dg"], Ho, Wo)).astype(np.float32)
    offset[:, :, 0, :] *= 8.0            # push some taps outside the image
    want = oracle.deform_im2col(data, offset, k, k, cfg["pad"], cfg["stride"], cfg["dil"], cfg["dg"])
    want_cl = want.reshape(N, C, k * k, Ho * Wo).transpose(0, 3, 2, 1).reshape(N, Ho * Wo, k * k * C)
    got = hip.deform_im2col_cl(t(np.ascontiguousarray(data.transpose(0, 2, 3, 1))),
                               t(np.ascontiguousarray(offset.transpose(0, 2, 3, 1))), k, k, cfg["pad"], cfg["stride"],
                               cfg["dil"], cfg["dg"]).cpu().numpy()
    np.testing.assert_array_equal(got, want_cl)


@pytest.mark.parametrize("shape", [(1, 64, 300, 500), (1, 256, 38, 63), (2, 7, 5, 3)])
def test_scale_shift_relu_bit_exact(hip, shape):
    rs = np.random.RandomState(shape[1])
    x = rs.randn(*shape).astype(np.float32)
    sc, sh = rs.rand(shape[1]).astype(np.float32) + 0.5, rs.randn(shape[1]).astype(np.float32)
    for relu in (True, False):
        want = oracle.scale_shift_relu(x, sc, sh, relu)
        got = hip.scale_shift_relu(t(x), t(sc), t(sh), relu).cpu().numpy()
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("shape", [(1, 64, 150, 250), (1, 1024, 5, 8), (2, 7, 5, 3)])
def test_scale_shift_leaky_bit_exact(hip, shape):
    rs = np.random.RandomState(shape[1] + 7)
    x = rs.randn(*shape).astype(np.float32)
    sc, sh = rs.rand(shape[1]).astype(np.float32) + 0.5, rs.randn(shape[1]).astype(np.float32)
    want = oracle.scale_shift_leaky(x, sc, sh, 0.1)
    got = hip.scale_shift_leaky(t(x), t(sc), t(sh), 0.1).cpu().numpy()
    np.testing.assert_array_equal(got, want)
    # == torch's bias add + leaky_relu (what the oracle graph does) when scale is 1
    ones = np.ones(shape[1], np.float32)
    ref = torch.nn.functional.leaky_relu(torch.from_numpy(x) + torch.from_numpy(sh).view(1, -1, 1, 1), 0.1).numpy()
    np.testing.assert_array_equal(hip.scale_shift_leaky(t(x), t(ones), t(sh), 0.1).cpu().numpy(), ref)


@pytest.mark.parametrize("shape", [(150 * 250, 64), (2394, 1024), (7, 4)])
def test_scale_shift_relu_channels_last_bit_exact(hip, shape):
    """(rows, C) maps with the channel fastest == the NCHW statement with N = rows, HW = 1."""
    rs = np.random.RandomState(shape[1])
    x = rs.randn(*shape).astype(np.float32)
    sc, sh = rs.rand(shape[1]).astype(np.float32) + 0.5, rs.randn(shape[1]).astype(np.float32)
    for relu in (True, False):
        want = oracle.scale_shift_relu(x.reshape(shape[0], shape[1], 1), sc, sh, relu).reshape(shape)
        got = hip.scale_shift_relu_cl(t(x), t(sc), t(sh), relu).cpu().numpy()
        np.testing.assert_array_equal(got, want)
    xt = t(x)
    hip.scale_shift_relu_cl(xt, t(sc), t(sh), True, out=xt)            # in place
    np.testing.assert_array_equal(xt.cpu().numpy(), want if relu else oracle.scale_shift_relu(x.reshape(shape[0], shape[1], 1), sc, sh, True).reshape(shape))
    with pytest.raises(hip.LsfaError):
        hip.scale_shift_relu_cl(torch.zeros(5, 6, device=DEV), torch.ones(6, device=DEV), torch.zeros(6, device=DEV))


# ------------------------------------------------------------------ profiling hook ----
def test_prof_hooks_report_launches(hip):
    hip.prof_enable(True)
    try:
        feat = torch.randn(1, 64, 38, 63, device=DEV)
        for _ in range(3):
            hip.warp_bilinear(feat, torch.zeros(1, 2, 38, 63, device=DEV))
        stats = hip.prof_read()
        assert stats["warp_bilinear"][1] == 3 and stats["warp_bilinear"][0] > 0
        assert stats["proposal"][1] == 0
    finally:
        hip.prof_enable(False)


# ------------------------------------------------------------------ compressed-domain MVs ----
def synthetic_block_mvs(rs, width, height, n_extra=40):
    """A decoder-like block list: the 16x16 macroblock grid with small motions (some zero, some leaving the
    frame), plus 8x8 / 16x8 / odd-sized partitions scattered on top (later entries overwrite earlier ones)."""
    rows = []
    for by in range(0, height, 16):
        for bx in range(0, width, 16):
            dx, dy = rs.randint(-9, 10), rs.randint(-9, 10)
            if rs.rand() < 0.2:
                dx = dy = 0
            rows.append([-1, 16, 16, bx + 8 - dx, by + 8 - dy, bx + 8, by + 8])
    for _ in range(n_extra):
        w, h = [(8, 8), (16, 8), (8, 16), (5, 7), (16, 16)][rs.randint(0, 5)]
        x, y = rs.randint(-4, width + 4), rs.randint(-4, height + 4)
        rows.append([-1, w, h, x - rs.randint(-20, 21), y - rs.randint(-20, 21), x, y])
    return np.array(rows, np.int32)


@pytest.mark.parametrize("size", [(96, 64), (1000, 600), (37, 23)])
def test_mv_accumulation_bit_exact(hip, size):
    """lsfa_mv_* vs the oracle's statement of coviar_data_loader.c:71-177 over a 5-frame GOP: the accumulated
    source map after every frame, the motion-vector field and the residual, all int32, all equal."""
    width, height = size
    rs = np.random.RandomState(width)
    acc = hip.MotionVectorAccumulator(width, height, DEV)
    want = oracle.coviar_identity(width, height)
    np.testing.assert_array_equal(acc.accu.cpu().numpy(), want)
    for frame in range(5):
        mvs = synthetic_block_mvs(rs, width, height)
        if frame == 3:
            mvs = mvs[:0]                                  # a frame without motion vectors
        acc.add_frame(torch.from_numpy(mvs))
        want = oracle.coviar_accumulate(mvs, want)
        np.testing.assert_array_equal(acc.accu.cpu().numpy(), want, err_msg="frame %d" % frame)
    assert (want != oracle.coviar_identity(width, height)).any()
    np.testing.assert_array_equal(acc.motion_vectors().cpu().numpy(), oracle.coviar_mv(want))
    cur = rs.randint(0, 256, (height, width, 3)).astype(np.uint8)
    ref = rs.randint(0, 256, (height, width, 3)).astype(np.uint8)
    np.testing.assert_array_equal(acc.residual(torch.from_numpy(cur), torch.from_numpy(ref)).cpu().numpy(),
                                  oracle.coviar_residual(cur, ref, want))
    # r5: from the decoder's blocks to the network's inputs without leaving the device (get_image, lib/utils/image.py:52-63: the motion vectors
    # negated, both maps through transform_mv_res)
    means, ps, sc = (102.9801, 115.9465, 122.7717), 1.0, 1.25
    want_mv, want_res = np_ref.transform_mv_res(-oracle.coviar_mv(want).astype(np.float32), oracle.coviar_residual(cur, ref, want).astype(np.float32), sc, means, ps)
    got_mv, got_res = acc.network_inputs(torch.from_numpy(cur), torch.from_numpy(ref), sc, means, ps)
    np.testing.assert_array_equal(got_mv.cpu().numpy(), want_mv.astype(np.float32))
    np.testing.assert_array_equal(got_res.cpu().numpy(), want_res.astype(np.float32))
    acc.reset()
    np.testing.assert_array_equal(acc.accu.cpu().numpy(), oracle.coviar_identity(width, height))


# ------------------------------------------------------------------ fp32 MFMA convolution ----
@pytest.mark.parametrize("cfg", [
    dict(N=1, H=38, W=63, Cin=256, Cout=256, k=3, stride=1, dil=1),      # stage-3 conv2 (taps split over 3 slices)
    dict(N=1, H=75, W=125, Cin=128, Cout=128, k=3, stride=1, dil=1),     # stage-2 conv2
    dict(N=1, H=75, W=125, Cin=256, Cout=256, k=3, stride=2, dil=1),     # stage3_unit1 (stride 2)
    dict(N=1, H=38, W=63, Cin=512, Cout=512, k=3, stride=1, dil=2),      # dilated
    dict(N=2, H=13, W=9, Cin=64, Cout=64, k=3, stride=1, dil=1),         # batch 2, ragged pixel tile
    dict(N=1, H=20, W=17, Cin=96, Cout=128, k=1, stride=1, dil=1),       # 1x1
])
def test_conv_nhwc_mfma_vs_torch(hip, cfg):
    """lsfa_conv_nhwc_fwd (fp32 MFMA implicit GEMM, bias + ReLU epilogue) vs torch's CPU convolution in float64:
    the fp32 result must sit within fp32 round-off of a K-term sum (both are correctly ordered sums of the same
    exact products; only the summation order differs), and a second launch must reproduce the first bit for bit
    (the tap slices are added in a fixed order)."""
    import torch.nn.functional as F
    rs = np.random.RandomState(cfg["Cin"] + cfg["H"])
    N, H, W, Cin, Cout, k, stride, dil = (cfg[x] for x in ("N", "H", "W", "Cin", "Cout", "k", "stride", "dil"))
    pad = dil * (k // 2)
    x = rs.randn(N, Cin, H, W).astype(np.float32)
    w = (rs.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rs.randn(Cout).astype(np.float32)
    want = torch.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                               stride, pad, dil)).permute(0, 2, 3, 1).numpy()
    xt = t(x).permute(0, 2, 3, 1).contiguous()
    wk = hip.conv_weight_kc(t(w))
    got = hip.conv_nhwc(xt, wk, t(b), k, k, stride, pad, dil, relu=True)
    again = hip.conv_nhwc(xt, wk, t(b), k, k, stride, pad, dil, relu=True)
    assert torch.equal(got, again)
    g = got.cpu().numpy()
    assert g.shape == want.shape
    scale = np.abs(want).max()
    assert np.abs(g - want).max() < 2e-6 * np.sqrt(Cin * k * k) * max(scale, 1.0)
    # no bias / no ReLU path
    want2 = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None, stride, pad, dil).permute(0, 2, 3, 1).numpy()
    g2 = hip.conv_nhwc(xt, wk, None, k, k, stride, pad, dil, relu=False).cpu().numpy()
    assert np.abs(g2 - want2).max() < 2e-6 * np.sqrt(Cin * k * k) * max(np.abs(want2).max(), 1.0)
    assert (g2 < 0).any()


def test_copy_many_is_one_launch_of_plain_copies(hip):
    """lsfa_copy_many: a frame's image, motion vectors and residual into the graph's static buffers as one launch."""
    rs = np.random.RandomState(3)
    srcs = [t(rs.randn(1, 3, 60, 100).astype(np.float32)), t(rs.randn(1, 2, 38, 63).astype(np.float32)),
            t(rs.randn(1, 3, 38, 63).astype(np.float32)), t(rs.randint(0, 99, (7,)).astype(np.int32))]
    dsts = [torch.zeros_like(x) for x in srcs]
    hip.copy_many(list(zip(dsts, srcs)))
    for d, s_ in zip(dsts, srcs):
        assert torch.equal(d, s_)
    hip.copy_many([(dsts[1], srcs[1] * 2)])
    assert torch.equal(dsts[1], srcs[1] * 2) and torch.equal(dsts[0], srcs[0])
    with pytest.raises(hip.LsfaError):
        hip.copy_many([(dsts[0], srcs[1])])
    with pytest.raises(hip.LsfaError):
        hip.copy_many([(dsts[0].double(), srcs[0].double())])
    # a segment's worth: 27 jobs of three sizes (one of them odd) into slices of batched buffers
    big = [torch.zeros((9, 3, 60, 100), device=DEV), torch.zeros((9, 2, 38, 63), device=DEV), torch.zeros((9, 3, 37, 21), device=DEV)]
    parts = [[t(rs.randn(1, *b.shape[1:]).astype(np.float32)) for _ in range(9)] for b in big]
    hip.copy_many([(b[f:f + 1], parts[k][f]) for f in range(9) for k, b in enumerate(big)])
    for k, b in enumerate(big):
        assert torch.equal(b, torch.cat(parts[k], 0))
    with pytest.raises(hip.LsfaError):
        hip.copy_many([(dsts[1], srcs[1])] * 33)


def test_copy_many_zero_fills_where_there_is_no_source(hip):
    """r5: a job without a source clears its destination (the frame path's amax slots, status words and padded maps are zeroed by this
    kernel, not by a PyTorch fill), alongside plain copies in the same launch; empty destinations are skipped."""
    rs = np.random.RandomState(5)
    a, b, c = t(rs.randn(3, 37, 21).astype(np.float32)), t(rs.randn(257).astype(np.float32)), t(rs.randint(1, 99, (1031,)).astype(np.int32))
    src = t(rs.randn(257).astype(np.float32))
    keep = a.clone()
    hip.copy_many([(b, src), (c, None), (torch.empty((0,), device=DEV), None)])
    assert torch.equal(b, src) and int(c.abs().sum()) == 0 and torch.equal(a, keep)
    hip.copy_many([(a, None)])
    assert float(a.abs().sum()) == 0.0 and a.shape == keep.shape
    z = hip.zeros_f32((2, 5, 7), DEV)
    assert z.dtype == torch.float32 and z.shape == (2, 5, 7) and float(z.abs().sum()) == 0.0
    hip.copy_many([])                                              # nothing to do: no launch, no error


@pytest.mark.parametrize("case", [(600, 1000, 1.0, (0.0, 0.0, 0.0), 1.0), (720, 1280, 0.78125, (102.9801, 115.9465, 122.7717), 1.0),
                                  (72, 128, 100.0 / 128, (3.0, 5.0, 7.0), 0.5), (37, 53, 1.7, (1.5, 2.5, 3.5), 0.017), (16, 16, 1.0, (0.0, 0.0, 0.0), 1.0),
                                  (5, 9, 0.6, (1.0, 2.0, 3.0), 2.0)])
def test_transform_mv_res_on_the_device_is_the_references(hip, case):
    """r5 (lsfa_transform_mv_res): decoded motion vectors + residual -> the network's stride-16 inputs in ONE launch, bit for bit
    the reference's arithmetic (oracle/np_ref.py::transform_mv_res, float64, rounded to float32 where the reference hands the arrays to the
    executor): first resize in float32, padding, the in-place channel loop and the 1 / 16 resize in float64.  int32 maps (what
    lsfa_mv_field / lsfa_mv_residual leave on the device) and float32 maps; odd sizes, up- and down-scaling, a one-cell output."""
    H, W, scale, means, ps = case
    rs = np.random.RandomState(H + W)
    mv = rs.randint(-40, 40, (H, W, 2)).astype(np.int32)
    res = rs.randint(-255, 256, (H, W, 3)).astype(np.int32)
    want_mv, want_res = np_ref.transform_mv_res(mv, res, scale, means, ps)
    got_mv, got_res = hip.transform_mv_res(t(mv), t(res), scale, means, ps)
    assert tuple(got_mv.shape) == want_mv.shape and tuple(got_res.shape) == want_res.shape
    np.testing.assert_array_equal(got_mv.cpu().numpy(), want_mv.astype(np.float32))
    np.testing.assert_array_equal(got_res.cpu().numpy(), want_res.astype(np.float32))
    # float32 maps with fractional values (an already resized or filtered field)
    mvf = (mv + rs.rand(H, W, 2)).astype(np.float32)
    resf = (res * 0.37).astype(np.float32)
    want_mv, want_res = np_ref.transform_mv_res(mvf, resf, scale, means, ps)
    got_mv, got_res = hip.transform_mv_res(t(mvf), t(resf), scale, means, ps)
    np.testing.assert_array_equal(got_mv.cpu().numpy(), want_mv.astype(np.float32))
    np.testing.assert_array_equal(got_res.cpu().numpy(), want_res.astype(np.float32))
    # the host-side module dispatches device maps to the same launch
    from lsfa_amd.utils import image
    h_mv, h_res = image.transform_mv_res(t(mv), t(res), scale, means, ps)
    assert h_mv.is_cuda and torch.equal(h_mv, hip.transform_mv_res(t(mv), t(res), scale, means, ps)[0])
    with pytest.raises(hip.LsfaError):
        hip.transform_mv_res(t(mv).double(), t(res).double(), scale, means, ps)
    # get_image negates the decoder's motion vectors before the transform (lib/utils/image.py:54)
    want_mv, _ = np_ref.transform_mv_res(-mv.astype(np.float32), res.astype(np.float32), scale, means, ps)
    got_mv, _ = hip.transform_mv_res(t(mv), t(res), scale, means, ps, negate_mv=True)
    np.testing.assert_array_equal(got_mv.cpu().numpy(), want_mv.astype(np.float32))


@pytest.mark.parametrize("case", [(600, 1000, 1.0, 0), (720, 1280, 0.78125, 16), (480, 640, 1.25, 32), (37, 53, 1.7, 16), (9, 7, 0.6, 0)])
def test_image_resize_transform_on_the_device_is_the_references(hip, case):
    """r5 (lsfa_image_resize_transform): a decoded frame -> `data` in one launch = resize (cv2's float INTER_LINEAR as restated in
    oracle/np_ref.py, zero padding to the image stride) + transform (float64, rounded to float32 at the executor), bit for bit; uint8 and
    float32 frames, two per launch; at scale 1 without padding it is lsfa_image_transform_u8."""
    H, W, scale, stride = case
    rs = np.random.RandomState(H * 3 + W)
    means, ps = (102.9801, 115.9465, 122.7717), 0.5
    ims = rs.randint(0, 256, (2, H, W, 3)).astype(np.uint8)

    def want_of(im):
        r = np_ref.cv2_resize_linear(im.astype(np.float32), scale, scale)
        if stride:
            ph, pw = -(-r.shape[0] // stride) * stride, -(-r.shape[1] // stride) * stride
            p = np.zeros((ph, pw, 3))            # float64, like the reference's np.zeros (image.py:291): transform then subtracts in float64
            p[:r.shape[0], :r.shape[1]] = r
            r = p
        return np_ref.transform(r, means, ps).astype(np.float32)

    want = np.concatenate([want_of(ims[0]), want_of(ims[1])], 0)
    got = hip.image_resize_transform(t(ims), scale, means, ps, stride=stride)
    assert tuple(got.shape) == want.shape
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    got_f = hip.image_resize_transform(t(ims.astype(np.float32)), scale, means, ps, stride=stride)
    np.testing.assert_array_equal(got_f.cpu().numpy(), want)
    if scale == 1.0 and stride == 0 and (H * W) % 4 == 0:         # zero means (the resnet-101 configuration): the uint8-image form of `transform` agrees
        assert torch.equal(hip.image_resize_transform(t(ims), 1.0, (0.0, 0.0, 0.0), ps), hip.image_transform_u8(t(ims), (0.0, 0.0, 0.0), ps))


def test_image_table_entry_points_equal_the_dense_ones(hip):
    """r6 (lsfa_avgpool_nchw_tbl, lsfa_stem_conv7x7s2_tbl, lsfa_ptr_table_set): N frames that live in separate tensors, read through a device
    table of per-image pointers, give the bits of the same frames stacked into one (N, 3, H, W) tensor - and a table rewritten to other
    frames gives theirs (what a replayed graph sees between two passes)."""
    rs = np.random.RandomState(77)
    N, H, W = 5, 37, 53
    frames = [t((rs.rand(3, H, W) * 255).astype(np.float32)) for _ in range(N)]
    pad = [torch.empty(1000 + 17 * i, device=DEV) for i in range(N)]                       # (keeps the frames apart in memory)
    dense = torch.stack(frames, 0)
    w = (rs.randn(64, 3, 7, 7) * 0.05).astype(np.float32)
    wl, b = hip.stem_weight_layout(t(w)), t(rs.randn(64).astype(np.float32))
    sc, sh = t(rs.uniform(0.01, 0.03, 3).astype(np.float32)), t(rs.uniform(-3, -1, 3).astype(np.float32))
    tbl = hip.ImageTable(N, 3, H, W, DEV).set(frames)
    for k in (2, 4):
        assert torch.equal(hip.avgpool_nchw(tbl, k), hip.avgpool_nchw(dense, k))
    slots = hip.amax_slots(2, DEV)
    y_t = hip.stem_conv(tbl, wl, b, sc, sh, amax_out=slots[0])
    y_d = hip.stem_conv(dense, wl, b, sc, sh, amax_out=slots[1])
    assert torch.equal(y_t, y_d) and torch.equal(slots[0], slots[1])
    other = [frames[(i + 2) % N] for i in range(N)]
    tbl.set(other)
    assert torch.equal(hip.stem_conv(tbl, wl, b, sc, sh), hip.stem_conv(torch.stack(other, 0), wl, b, sc, sh))
    assert torch.equal(hip.avgpool_nchw(tbl, 4), hip.avgpool_nchw(torch.stack(other, 0), 4))
    with pytest.raises(hip.LsfaError):
        tbl.set(frames[:-1])
    with pytest.raises(hip.LsfaError):
        tbl.set([f.double() for f in frames])
    del pad


@pytest.mark.parametrize("case", [(600, 1000, 1.0, 0), (720, 1280, 0.78125, 0), (480, 640, 1.25, 32), (37, 53, 1.7, 16), (9, 7, 0.6, 0)])
def test_image_resize_transform_u8_fixed_point_path(hip, case):
    """r6 (is_u8 = 2): the last frame of a video reaches `resize` as cv2.imread's uint8 image (lib/utils/image.py:45) and OpenCV interpolates it in
    fixed point; oracle/np_ref.py::cv2_resize_linear_u8 restates that path (parity unpinned) and the kernel equals it + `transform`'s uint8 rule
    (float64 subtraction) bit for bit, padding included; at scale 1 it is lsfa_image_transform_u8."""
    H, W, scale, stride = case
    rs = np.random.RandomState(H + 5 * W)
    means, ps = (102.9801, 115.9465, 122.7717), 0.5
    ims = rs.randint(0, 256, (2, H, W, 3)).astype(np.uint8)

    def want_of(im):
        r = np_ref.cv2_resize_linear_u8(im, scale, scale)
        if stride:
            ph, pw = -(-r.shape[0] // stride) * stride, -(-r.shape[1] // stride) * stride
            p = np.zeros((ph, pw, 3))
            p[:r.shape[0], :r.shape[1]] = r
            r = p
        return np_ref.transform(r, means, ps).astype(np.float32)

    want = np.concatenate([want_of(ims[0]), want_of(ims[1])], 0)
    got = hip.image_resize_transform(t(ims), scale, means, ps, stride=stride, u8_fixed_point=True)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    floaty = hip.image_resize_transform(t(ims), scale, means, ps, stride=stride).cpu().numpy()
    assert np.abs(floaty - got.cpu().numpy()).max() <= 1.25 * ps               # about one intensity level from the float interpolation (final rounding 0.5 + two truncations 0.5 + 11-bit coefficients)
    if scale == 1.0 and (H * W) % 4 == 0:
        assert torch.equal(got, hip.image_transform_u8(t(ims), means, ps))


def test_image_resize_transform_golden_g6_float32_frame(hip, golden):
    """... and against G6: the reference's own `transform` on a float32 frame with list means (float32 subtraction, float64 product)."""
    imf = golden["g6_im"].astype(np.float32) * np.float32(0.731)
    got = hip.image_resize_transform(t(imf), 1.0, [103.94, 116.78, 123.68], 0.017)
    np.testing.assert_array_equal(got.cpu().numpy(), golden["g6_transform_f32_list_means"].astype(np.float32))


def test_image_resize_transform_golden_g6_padded_frame(hip, golden):
    """... and G6's padded case: the reference's own resize(stride=16) + transform on a float32 frame - the padded copy is a float64
    image, so the mean is subtracted in float64."""
    big = golden["g6_resize_in"].astype(np.float32)
    got = hip.image_resize_transform(t(big), float(golden["g6_resize_scale"]), [103.94, 116.78, 123.68], 0.017, stride=16)
    np.testing.assert_array_equal(got.cpu().numpy(), golden["g6_resize_f32_stride16_transform"].astype(np.float32))


def test_transform_mv_res_golden_g6_on_the_device(hip, golden):
    """... and against G6: the reference's own transform_mv_res run around the restated INTER_LINEAR (tests/golden/make_golden.py)."""
    mv, res, means, ps = golden["g6_mv"], golden["g6_res"], golden["g6_means"], float(golden["g6_pixel_scale"])
    for tag, sc in (("s1", 1.0), ("s16", 1.6)):
        got_mv, got_res = hip.transform_mv_res(t(np.ascontiguousarray(mv)), t(np.ascontiguousarray(res).astype(mv.dtype)), sc, means, ps)
        np.testing.assert_array_equal(got_mv.cpu().numpy(), golden["g6_mv_tensor_" + tag].astype(np.float32))
        np.testing.assert_array_equal(got_res.cpu().numpy(), golden["g6_res_tensor_" + tag].astype(np.float32))


# ------------------------------------------------------------------ ResNet stem ----
@pytest.mark.parametrize("shape", [(1, 600, 1000), (1, 150, 250), (2, 37, 50), (1, 9, 8), (3, 71, 131)])
def test_stem_conv_pool_vs_torch(hip, shape):
    """bn_data + conv0 (7x7/2, pad 3, bn0 folded) + ReLU as one launch, then pool0 (3x3/2 max, pad 1): against torch in
    float64 (two fp16 pieces per operand, three matrix products, fp32 accumulation: the error of an fp32 convolution of
    147 terms, 2e-6*sqrt(147) of the output scale) and, for the pooling, exactly."""
    import torch.nn.functional as F
    N, H, W = shape
    rs = np.random.RandomState(H + W)
    x = (rs.rand(N, 3, H, W) * 255).astype(np.float32)
    w = (rs.randn(64, 3, 7, 7) * 0.05).astype(np.float32)
    b = rs.randn(64).astype(np.float32)
    sc, sh = rs.uniform(0.01, 0.03, 3).astype(np.float32), rs.uniform(-3, -1, 3).astype(np.float32)
    xa = torch.from_numpy(x).double() * torch.from_numpy(sc).double().view(1, 3, 1, 1) + torch.from_numpy(sh).double().view(1, 3, 1, 1)
    want = torch.relu(F.conv2d(xa, torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=2, padding=3))
    wl = hip.stem_weight_layout(t(w))
    got = hip.stem_conv(t(x), wl, t(b), t(sc), t(sh))
    assert got.shape == (N, want.shape[2], want.shape[3], 64)
    g = got.permute(0, 3, 1, 2).cpu().double()
    assert float((g - want).abs().max()) < 2e-6 * np.sqrt(147) * max(float(want.abs().max()), 1.0)
    # without bn_data
    want0 = torch.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None, stride=2, padding=3))
    got0 = hip.stem_conv(t(x), wl, None).permute(0, 3, 1, 2).cpu().double()
    assert float((got0 - want0).abs().max()) < 2e-6 * np.sqrt(147) * max(float(want0.abs().max()), 1.0)
    pooled = hip.maxpool3x3s2_nhwc(got)
    ref = F.max_pool2d(got.permute(0, 3, 1, 2), 3, 2, 1)
    assert torch.equal(pooled.permute(0, 3, 1, 2), ref)
    # with the first unit's bn1 + relu1 as a second output of the same launch
    s2, t2 = t(rs.uniform(0.5, 1.5, 64).astype(np.float32)), t(rs.randn(64).astype(np.float32))
    p1, p2 = hip.maxpool3x3s2_nhwc(got, scale2=s2, shift2=t2)
    assert torch.equal(p1, pooled) and torch.equal(p2, torch.relu(pooled * s2 + t2))


def test_stem_conv_batch_invariance_accumulation_and_amax(hip):
    """What the pipeline relies on beyond the values: (a) an image's result is the same bits alone and inside a batch, whatever the
    other images hold (the input's fp16 scale is taken per tile from the tile's own patch; the batched passes are compared with the
    frame-by-frame ones bit for bit); (b) FlowNet's flow_conv1 = two passes, the second accumulating into the first in place with
    LeakyReLU(0.1) (resnet_v1_101_flownet_rfcn.py:153), against the 6-channel convolution in float64; (c) amax_out receives max|y|
    exactly; (d) weights and inputs of very different magnitudes per channel keep fp32 accuracy relative to each OUTPUT channel's
    scale (the weight scale is per output channel)."""
    import torch.nn.functional as F
    rs = np.random.RandomState(77)
    H, W = 53, 141
    x = (rs.rand(4, 3, H, W) * 255).astype(np.float32)
    x[1] *= 1e-3                                    # a dark image next to bright ones
    x[2, :, 20:30, 50:90] = 6.0e4                   # a saturated patch
    w = (rs.randn(64, 3, 7, 7) * 0.05).astype(np.float32)
    w *= np.exp(rs.uniform(-6, 6, 64)).astype(np.float32)[:, None, None, None]     # channels 5 decades apart
    b = rs.randn(64).astype(np.float32)
    wl = hip.stem_weight_layout(t(w))
    am = hip.amax_slots(1, DEV)[0]
    both = hip.stem_conv(t(x), wl, t(b), act=0, amax_out=am)
    for i in range(4):
        alone = hip.stem_conv(t(x[i:i + 1]), wl, t(b), act=0)
        assert torch.equal(alone[0], both[i]), i
    assert float(am.view(torch.float32).max()) == float(both.abs().max())
    want = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(), stride=2, padding=3)
    err = (both.permute(0, 3, 1, 2).cpu().double() - want).abs().amax(dim=(2, 3))          # (image, channel)
    scale = want.abs().amax(dim=(2, 3)).clamp_min(1e-30)
    assert float((err / scale).max()) < 2e-6 * np.sqrt(147), float((err / scale).max())
    # two passes = the 6-channel convolution
    x2 = (rs.rand(2, 3, H, W)).astype(np.float32)
    x1 = (rs.rand(2, 3, H, W)).astype(np.float32)
    w6 = (rs.randn(64, 6, 7, 7) * 0.05).astype(np.float32)
    r1 = hip.stem_conv(t(x1), hip.stem_weight_layout(t(w6[:, 0:3].copy())), None, act=0)
    r1 = hip.stem_conv(t(x2), hip.stem_weight_layout(t(w6[:, 3:6].copy())), t(b), out=r1, accum=r1, act=2)
    want6 = F.leaky_relu(F.conv2d(torch.from_numpy(np.concatenate([x1, x2], 1)).double(), torch.from_numpy(w6).double(),
                                  torch.from_numpy(b).double(), stride=2, padding=3), 0.1)
    assert float((r1.permute(0, 3, 1, 2).cpu().double() - want6).abs().max()) < 2e-6 * np.sqrt(294) * float(want6.abs().max())
    with pytest.raises(hip.LsfaError):
        hip.stem_conv(t(x1), t(w6[:, 0:3].transpose(1, 2, 3, 0).copy()), None)          # r3's float layout is not the fragments


def test_image_transform_u8_is_the_references_transform(hip, golden):
    """lsfa_image_transform_u8 (r5: the frame's upload path) against G6 - the output of the reference's own `transform`
    (lib/utils/image.py:296-308, run by tests/golden/make_golden.py) rounded to float32 where the reference hands it to the executor:
    bit-exact, with real pixel means / scale and with the resnet-101 configuration's zero means; a batch of frames; and the oracle's
    restatement at the benchmark's frame size."""
    from oracle import np_ref
    im = golden["g6_im"]
    means, ps = golden["g6_means"], float(golden["g6_pixel_scale"])
    dev_im = torch.from_numpy(im)[None].to(DEV)
    out = hip.image_transform_u8(dev_im, means, ps)
    assert np.array_equal(out.cpu().numpy(), golden["g6_transform"].astype(np.float32))
    out0 = hip.image_transform_u8(dev_im, (0.0, 0.0, 0.0), 1.0)
    assert np.array_equal(out0.cpu().numpy(), golden["g6_transform_zero_means"].astype(np.float32))
    rs = np.random.RandomState(5)
    batch = rs.randint(0, 256, (3, 600, 1000, 3)).astype(np.uint8)
    got = hip.image_transform_u8(torch.from_numpy(batch).to(DEV), means, ps).cpu().numpy()
    for n in range(3):
        assert np.array_equal(got[n:n + 1], np_ref.transform(batch[n], means, ps).astype(np.float32))
    with pytest.raises(hip.LsfaError):
        hip.image_transform_u8(torch.zeros((1, 3, 7, 3), dtype=torch.uint8, device=DEV))      # H*W % 4 != 0


@pytest.mark.parametrize("shape,k", [((1, 3, 600, 1000), 4), ((2, 3, 37, 50), 4), ((1, 5, 9, 7), 2)])
def test_avgpool_nchw_vs_torch_ceil_mode(hip, shape, k):
    """mx Pooling(avg, pooling_convention='full') = torch avg_pool2d(ceil_mode=True) without padding: edge windows are
    clipped to the image and divided by what they hold."""
    import torch.nn.functional as F
    x = t(np.random.RandomState(shape[2]).rand(*shape).astype(np.float32) * 255)
    got = hip.avgpool_nchw(x, k)
    want = F.avg_pool2d(x.double(), k, k, ceil_mode=True)
    assert got.shape == want.shape
    assert float((got.double() - want).abs().max()) < 1e-4


# ------------------------------------------------------------------ split-bf16 convolution ----
SPLIT_CFGS = [
    dict(N=1, H=38, W=63, Cin=256, Cout=256, k=3, stride=1, dil=1),      # res4 conv2 (K cut into slices)
    dict(N=1, H=75, W=125, Cin=128, Cout=128, k=3, stride=1, dil=1),     # res3 conv2
    dict(N=1, H=75, W=125, Cin=256, Cout=256, k=3, stride=2, dil=1),     # stride 2
    dict(N=1, H=38, W=63, Cin=512, Cout=512, k=3, stride=1, dil=2),      # dilated
    dict(N=2, H=13, W=9, Cin=64, Cout=64, k=3, stride=1, dil=1),         # batch 2, ragged pixel tile, one chunk per tap
    dict(N=1, H=20, W=17, Cin=96, Cout=128, k=1, stride=1, dil=1),       # 1x1, odd chunk count
    dict(N=1, H=38, W=63, Cin=256, Cout=1024, k=1, stride=1, dil=1),     # res4 conv3 shape
    dict(N=1, H=38, W=63, Cin=256, Cout=1024, k=3, stride=1, dil=1),     # the fuse convolution
    dict(N=1, H=30, W=70, Cin=192, Cout=1024, k=3, stride=1, dil=2),     # dilation 2, 6 chunks per tap
    dict(N=2, H=20, W=63, Cin=256, Cout=1024, k=3, stride=1, dil=1),     # two images
    dict(N=1, H=21, W=40, Cin=256, Cout=64, k=3, stride=1, dil=6),       # feat_conv_3x3's dilation (general kernel)
    dict(N=1, H=38, W=63, Cin=512, Cout=1024, k=3, stride=1, dil=6),     # 304 workgroups: K cut into 5 by the rounds model
]


@pytest.mark.parametrize("pieces", [3, 2, 1])
@pytest.mark.parametrize("cfg", SPLIT_CFGS)
def test_conv_split_vs_float64_and_fp32_mfma(hip, cfg, pieces):
    """lsfa_conv_fwd: fp32 operands cut into three bf16 pieces (six partial products), two fp16 pieces (three) or one bf16
    piece (the bf16 mode) on the matrix pipe, fp32 accumulation.  Against a float64 convolution the fp32 forms must be inside
    the same bound as the fp32-MFMA kernel (2e-6 * sqrt(K) of the output scale) and not worse than 1.5x that kernel's own
    error; the bf16 form inside bf16 round-off; a second launch reproduces the first bit for bit (slices are added in a
    fixed order).  The epilogue's amax_out is the exact maximum of the output and the status word stays clear."""
    import torch.nn.functional as F
    rs = np.random.RandomState(cfg["Cin"] + cfg["H"] + 1)
    N, H, W, Cin, Cout, k, stride, dil = (cfg[x] for x in ("N", "H", "W", "Cin", "Cout", "k", "stride", "dil"))
    pad = dil * (k // 2)
    # activations like the network's: non-negative with a per-channel gain, a few exact zeros and tiny values
    x = (np.maximum(rs.randn(N, Cin, H, W), 0) * rs.uniform(0.1, 3.0, (1, Cin, 1, 1))).astype(np.float32)
    x[0, :, 0, 0] = 1e-30
    w = (rs.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)).astype(np.float32)
    b = rs.randn(Cout).astype(np.float32)
    want = torch.relu(F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                               stride, pad, dil)).permute(0, 2, 3, 1).numpy()
    xt = t(x).permute(0, 2, 3, 1).contiguous()
    sw = hip.SplitWeight(t(w), pieces=pieces)
    slots, status = hip.amax_slots(1, DEV)[0], hip.new_status(DEV)
    got = hip.conv_split(xt, sw, t(b), stride, pad, dil, relu=True, amax_out=slots, status=status)
    again = hip.conv_split(xt, sw, t(b), stride, pad, dil, relu=True)
    assert torch.equal(got, again)
    hip.check_status(status)
    assert slots.view(torch.float32).max().item() == got.abs().max().item()
    g = got.cpu().numpy()
    assert g.shape == want.shape
    scale = max(np.abs(want).max(), 1.0)
    err_split = np.abs(g - want).max()
    if pieces == 1:
        assert err_split < 1e-2 * scale, err_split         # one bf16 product per fp32 product: 2^-8 per term
        return
    assert err_split < 2e-6 * np.sqrt(Cin * k * k) * scale
    ref32 = hip.conv_nhwc(xt, hip.conv_weight_kc(t(w)), t(b), k, k, stride, pad, dil, relu=True).cpu().numpy()
    err_mfma = np.abs(ref32 - want).max()
    assert err_split < 1.5 * err_mfma + 1e-7 * scale, (err_split, err_mfma)
    rms = lambda a: float(np.sqrt(np.mean((a - want) ** 2)))
    assert rms(g) < 1.5 * rms(ref32) + 1e-8 * scale
    # no bias / no ReLU
    want2 = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None, stride, pad, dil).permute(0, 2, 3, 1).numpy()
    slots.zero_()
    g2t = hip.conv_split(xt, sw, None, stride, pad, dil, relu=False, amax_out=slots)
    g2 = g2t.cpu().numpy()
    assert np.abs(g2 - want2).max() < 2e-6 * np.sqrt(Cin * k * k) * max(np.abs(want2).max(), 1.0)
    assert (g2 < 0).any()
    assert slots.view(torch.float32).max().item() == g2t.abs().max().item()        # |negative| values count


def test_conv_split_cut_is_exact_and_products_are_fp32_grade(hip):
    """With one non-zero input channel the convolution is a single product per output: a*b from the six partial products
    must agree with the float64 product to 2^-22 relative (the three dropped terms are below 2^-23), for values whose low
    mantissa bits are all set, for negative, tiny and huge ones."""
    rs = np.random.RandomState(5)
    Cin, Cout, H, W = 32, 64, 4, 8
    vals = np.array([1.0 + 2.0 ** -23, 1.9999999, -3.1415927, 1e-20, -7e19, 0.33333334, 123456.79, -2.0 ** -100], np.float32)
    x = np.zeros((1, Cin, H, W), np.float32)
    x[0, 7] = np.resize(vals, H * W).reshape(H, W)
    w = np.zeros((Cout, Cin, 1, 1), np.float32)
    w[:, 7, 0, 0] = (rs.randn(Cout) * np.float32(1.7)).astype(np.float32) + np.float32(2.0 ** -12)
    got = hip.conv_split(t(x).permute(0, 2, 3, 1).contiguous(), hip.SplitWeight(t(w)), None).cpu().numpy()      # (1,H,W,Cout)
    want = x[0, 7].astype(np.float64)[:, :, None] * w[:, 7, 0, 0].astype(np.float64)[None, None, :]
    rel = np.abs(got[0] - want) / np.abs(want)
    assert rel.max() < 2.0 ** -22, rel.max()


@pytest.mark.parametrize("cfg", [dict(N=1, H=38, W=63, Cin=256, Cout=1024, k=3), dict(N=2, H=9, W=35, Cin=64, Cout=64, k=3),
                                 dict(N=1, H=38, W=63, Cin=256, Cout=256, k=3), dict(N=1, H=20, W=17, Cin=96, Cout=128, k=1)])
def test_conv_split_nchw_output_equals_nhwc_output(hip, cfg):
    """y_nchw: the same numbers in the reference operators' layout (the small net's fuse convolution feeds
    lsfa_warp_bilinear's `add` operand without a transposing copy), incl. residual and second output, for the ring kernel
    and the sliced path."""
    rs = np.random.RandomState(cfg["Cout"] + cfg["H"])
    N, H, W, Cin, Cout, k = (cfg[x] for x in ("N", "H", "W", "Cin", "Cout", "k"))
    x = t(rs.randn(N, H, W, Cin).astype(np.float32))
    sw = hip.SplitWeight(t((rs.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)).astype(np.float32)))
    b = t(rs.randn(Cout).astype(np.float32))
    res = t(rs.randn(N, H, W, Cout).astype(np.float32))
    sc2, sh2 = t(rs.uniform(0.5, 1.5, Cout).astype(np.float32)), t(rs.randn(Cout).astype(np.float32))
    y, y2 = hip.conv_split(x, sw, b, 1, k // 2, 1, relu=False, residual=res, out2=torch.empty_like(res), scale2=sc2, shift2=sh2)
    res_n = res.permute(0, 3, 1, 2).contiguous()
    z, z2 = hip.conv_split(x, sw, b, 1, k // 2, 1, relu=False, residual=res_n, out2=torch.empty_like(res_n), scale2=sc2, shift2=sh2,
                           nchw=True)
    assert z.shape == (N, Cout, H, W)
    assert torch.equal(z, y.permute(0, 3, 1, 2))
    assert torch.equal(z2, y2.permute(0, 3, 1, 2))


@pytest.mark.parametrize("pieces", [2, 1])
def test_conv_input_activation_at_the_cut(hip, pieces):
    """A pre-activation ResNet unit's bn1 + relu1 (dff_rfcn/symbols/resnet.py:78-80) applied where conv1 cuts its operand instead of
    being stored by the previous conv3 and read back: lsfa_conv_fwd with in_scale / in_shift on the raw sum == the same convolution
    on the stored max(x*s + t, 0), BIT FOR BIT, under every plan of the ring kernel (tile width, ring depth, wave roles, K slices),
    with stride 2 (a stage's first shortcut), a partial last tile and two images; and the producer's side: scale2 / shift2 without
    out2 store nothing but publish the same maximum and the same first output."""
    g = torch.Generator(device=DEV).manual_seed(70 + pieces)
    for (N, H, W, ci, co, stride) in ((2, 13, 23, 256, 64, 1), (1, 19, 31, 1024, 256, 1), (2, 14, 22, 512, 128, 2)):
        x = torch.randn((N, H, W, ci), device=DEV, generator=g) * 3.0
        sc, sh = torch.rand(ci, device=DEV, generator=g) + 0.5, torch.randn(ci, device=DEV, generator=g)
        sw = hip.SplitWeight(torch.randn((co, ci, 1, 1), device=DEV, generator=g) * 0.05, pieces=pieces)
        b = torch.randn(co, device=DEV, generator=g)
        act = torch.relu(x * sc + sh)                       # two roundings, like the epilogue that used to store it
        am = hip.amax_partial(act)
        plans = [(0, 0, 0, 0)] + [(kern, nt, st, sl) for kern in (1, 2) for nt in (2, 4) for st in (2, 3, 4) for sl in (1, 3)
                                  if not (nt == 4 and co % 128)]
        try:
            for kern, nt, st, sl in plans:
                hip.conv_plan_override(kernel=kern, nt=nt, st=st, slices=sl)
                s1, s2 = hip.amax_slots(2, DEV)
                want = hip.conv_split(act, sw, b, stride, 0, 1, relu=True, amax_in=am, amax_out=s1)
                got = hip.conv_split(x, sw, b, stride, 0, 1, relu=True, amax_in=am, amax_out=s2, in_scale=sc, in_shift=sh)
                if kern == 0:       # the plan's own choice: the stored map may go to the direct kernel (another summation order), the cut-time form never does
                    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()), (ci, co, stride)
                    continue
                assert torch.equal(got, want), (ci, co, stride, kern, nt, st, sl)
                assert torch.equal(s1.max(), s2.max())
        finally:
            hip.conv_plan_override()
    # the producer: second output not stored
    x = torch.randn((2, 13, 23, 64), device=DEV, generator=g)
    sw = hip.SplitWeight(torch.randn((256, 64, 1, 1), device=DEV, generator=g) * 0.05, pieces=pieces)
    res = torch.randn((2, 13, 23, 256), device=DEV, generator=g)
    sc2, sh2 = torch.rand(256, device=DEV, generator=g) + 0.5, torch.randn(256, device=DEV, generator=g)
    am = hip.amax_partial(x)
    try:
        for kern, nt, st, sl in ((0, 0, 0, 0), (1, 2, 2, 1), (1, 4, 2, 1), (2, 4, 3, 1), (1, 2, 3, 2), (2, 2, 2, 2)):
            hip.conv_plan_override(kernel=kern, nt=nt, st=st, slices=sl)
            s1, s2 = hip.amax_slots(2, DEV)
            y, y2 = hip.conv_split(x, sw, None, residual=res, out2=torch.empty_like(res), scale2=sc2, shift2=sh2, amax_in=am, amax_out=s1)
            z = hip.conv_split(x, sw, None, residual=res, scale2=sc2, shift2=sh2, amax_in=am, amax_out=s2)
            assert torch.equal(z, y) and torch.equal(s1.max(), s2.max()), (kern, nt, st, sl)
            assert s2.view(torch.float32).max().item() == y2.max().item()
    finally:
        hip.conv_plan_override()
    with pytest.raises(hip.LsfaError):      # nothing to publish into
        hip.conv_split(x, sw, None, residual=res, scale2=sc2, shift2=sh2, amax_in=am)
    with pytest.raises(hip.LsfaError):      # a 3x3 has padding: its zeros are not max(0*s + t, 0)
        sw3 = hip.SplitWeight(torch.randn((64, 64, 3, 3), device=DEV, generator=g) * 0.05, pieces=pieces)
        hip.conv_split(x, sw3, None, 1, 1, 1, amax_in=am, in_scale=sc2[:64].contiguous(), in_shift=sh2[:64].contiguous())


@pytest.mark.parametrize("pieces", [2, 3, 1])
def test_conv_ring_nchw_epilogue_through_lds(hip, pieces):
    """The unsliced ring kernel writes NCHW outputs as 128-byte runs of a plane (tiles turned in LDS, column-major with padded columns)
    instead of 4-byte pieces of 32 planes per instruction: under every tile width / ring depth / wave-role plan (K cut forced to one
    slice), with bias + residual + ReLU-free second output and with LeakyReLU alone, two images whose 128-pixel tiles straddle the image
    boundary and end in a partial tile - the same numbers as the channels-last output, transposed, and the same amax slots."""
    g = torch.Generator(device=DEV).manual_seed(40 + pieces)
    N, H, W, ci, co, k = 2, 13, 23, 64, 256, 3                    # 598 pixels: tiles of 128 straddle the images, the last one is partial
    x = torch.randn((N, H, W, ci), device=DEV, generator=g)
    sw = hip.SplitWeight(torch.randn((co, ci, k, k), device=DEV, generator=g) * 0.05, pieces=pieces)
    b = torch.randn(co, device=DEV, generator=g)
    res = torch.randn((N, H, W, co), device=DEV, generator=g)
    res_n = res.permute(0, 3, 1, 2).contiguous()
    sc2, sh2 = torch.rand(co, device=DEV, generator=g) + 0.5, torch.randn(co, device=DEV, generator=g)
    am = hip.amax_partial(x)
    try:
        for kern in (1, 2):
            for nt in (2, 4):
                for st in (2, 3):
                    hip.conv_plan_override(kernel=kern, nt=nt, st=st, slices=1)
                    s1, s2 = hip.amax_slots(2, DEV)
                    y, y2 = hip.conv_split(x, sw, b, 1, 1, 1, residual=res, out2=torch.empty_like(res), scale2=sc2, shift2=sh2, amax_in=am, amax_out=s1)
                    z, z2 = hip.conv_split(x, sw, b, 1, 1, 1, residual=res_n, out2=torch.empty_like(res_n), scale2=sc2, shift2=sh2, nchw=True,
                                           amax_in=am, amax_out=s2)
                    assert torch.equal(z, y.permute(0, 3, 1, 2)) and torch.equal(z2, y2.permute(0, 3, 1, 2)), (kern, nt, st)
                    assert s1.max().item() == s2.max().item() and s2.view(torch.float32).max().item() == z2.max().item()
                    yl = hip.conv_split(x, sw, b, 1, 1, 1, act=2, amax_in=am)
                    zl = hip.conv_split(x, sw, b, 1, 1, 1, act=2, nchw=True, amax_in=am)
                    assert torch.equal(zl, yl.permute(0, 3, 1, 2)), (kern, nt, st)
    finally:
        hip.conv_plan_override()
    # an overflowing scale is flagged through this epilogue too
    if pieces == 2:
        status = hip.new_status(DEV)
        try:
            hip.conv_plan_override(kernel=2, nt=4, st=3, slices=1)
            hip.conv_split(x * 100.0, sw, b, 1, 1, 1, relu=True, nchw=True, amax_in=am, status=status)
        finally:
            hip.conv_plan_override()
        with pytest.raises(hip.LsfaError, match="non-finite"):
            hip.check_status(status)


def test_conv_split_fused_tail_and_errors(hip):
    """residual add in place + second output (next unit's bn1 + ReLU), as lsfa_conv_nhwc_fused_fwd; shape errors."""
    import torch.nn.functional as F
    rs = np.random.RandomState(9)
    H, W, Cin, Cout = 38, 63, 256, 1024
    x = rs.randn(1, Cin, H, W).astype(np.float32)
    w = (rs.randn(Cout, Cin, 1, 1) / 16).astype(np.float32)
    res = rs.randn(1, Cout, H, W).astype(np.float32)
    sc2, sh2 = rs.uniform(0.5, 1.5, Cout).astype(np.float32), rs.randn(Cout).astype(np.float32)
    y64 = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double()) + torch.from_numpy(res).double()
    want = y64.permute(0, 2, 3, 1).numpy()
    buf = t(res).permute(0, 2, 3, 1).contiguous()
    out2 = torch.empty_like(buf)
    got, got2 = hip.conv_split(t(x).permute(0, 2, 3, 1).contiguous(), hip.SplitWeight(t(w)), None, out=buf, residual=buf, out2=out2,
                               scale2=t(sc2), shift2=t(sh2))
    assert got.data_ptr() == buf.data_ptr()
    assert np.abs(got.cpu().numpy() - want).max() < 2e-6 * 16 * max(np.abs(want).max(), 1.0)
    assert torch.equal(got2, torch.relu(got * t(sc2) + t(sh2)))
    with pytest.raises(hip.LsfaError):
        hip.SplitWeight(t(rs.randn(64, 48, 3, 3).astype(np.float32)))          # Cin % 32
    with pytest.raises(hip.LsfaError):
        hip.SplitWeight(t(rs.randn(96, 64, 3, 3).astype(np.float32)))          # Cout % 64
    with pytest.raises(hip.LsfaError):
        hip.conv_split(t(rs.randn(1, 8, 8, 32).astype(np.float32)), hip.SplitWeight(t(rs.randn(64, 64, 1, 1).astype(np.float32))))


@pytest.mark.parametrize("cfg", [dict(H=38, W=63, Cin=256, Cout=1024, k=1), dict(H=75, W=125, Cin=128, Cout=512, k=1),
                                 dict(H=19, W=11, Cin=64, Cout=64, k=3)])
def test_conv_nhwc_fused_residual_and_next_bn(hip, cfg):
    """lsfa_conv_nhwc_fused_fwd: conv3 of a pre-activation unit + the shortcut add IN PLACE + the next unit's
    bn1 + ReLU as a second output, vs the same three steps in float64."""
    import torch.nn.functional as F
    rs = np.random.RandomState(cfg["Cout"])
    H, W, Cin, Cout, k = (cfg[x] for x in ("H", "W", "Cin", "Cout", "k"))
    pad = k // 2
    x = rs.randn(1, Cin, H, W).astype(np.float32)
    w = (rs.randn(Cout, Cin, k, k) / np.sqrt(Cin * k * k)).astype(np.float32)
    res = rs.randn(1, Cout, H, W).astype(np.float32)
    sc2, sh2 = rs.uniform(0.5, 1.5, Cout).astype(np.float32), rs.randn(Cout).astype(np.float32)
    y64 = F.conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), None, 1, pad) + torch.from_numpy(res).double()
    y2_64 = torch.relu(y64 * torch.from_numpy(sc2).double().view(1, -1, 1, 1) + torch.from_numpy(sh2).double().view(1, -1, 1, 1))
    want, want2 = y64.permute(0, 2, 3, 1).numpy(), y2_64.permute(0, 2, 3, 1).numpy()
    xt = t(x).permute(0, 2, 3, 1).contiguous()
    buf = t(res).permute(0, 2, 3, 1).contiguous()               # residual in, sum out: the same buffer
    out2 = torch.empty_like(buf)
    got, got2 = hip.conv_nhwc(xt, hip.conv_weight_kc(t(w)), None, k, k, 1, pad, 1, relu=False, out=buf, residual=buf, out2=out2,
                              scale2=t(sc2), shift2=t(sh2))
    assert got.data_ptr() == buf.data_ptr()
    tol = 2e-6 * np.sqrt(Cin * k * k) * max(np.abs(want).max(), 1.0)
    assert np.abs(got.cpu().numpy() - want).max() < tol
    assert np.abs(got2.cpu().numpy() - want2).max() < 2 * tol
    # the second output is exactly the fp32 bn + relu of the first
    again = torch.relu(got * t(sc2) + t(sh2))
    assert torch.equal(got2, again)


def test_conv_split_views_and_transposed_convolution_phases(hip):
    """lsfa_conv_split_view_fwd: input read from the leading channels of a wider map, output written into a channel slice of a
    wider map, LeakyReLU; and Deconvolution(4x4, stride 2) + Crop(offset 1) (resnet_v1_101_flownet_rfcn.py:170-176) as four
    2x2-tap launches, one per output parity, against conv_transpose2d in float64."""
    import torch.nn.functional as F
    torch.manual_seed(3)
    dev = DEV
    # --- strided 5x5 convolution between views (FlowNet conv3: 128 -> 256, stride 2, pad 2) ---
    N, H, W, Cin, Cout, L, Lout, c0 = 1, 19, 32, 128, 256, 160, 416, 64
    xw = torch.randn(N, H, W, L, device=dev)
    w = torch.randn(Cout, Cin, 5, 5, device=dev) * 0.02
    b = torch.randn(Cout, device=dev)
    sw = hip.SplitWeight(w)
    Ho, Wo = (H + 4 - 5) // 2 + 1, (W + 4 - 5) // 2 + 1
    out = torch.full((N, Ho, Wo, Lout), 7.0, device=dev)
    hip.conv_split_view(xw, sw, b, out, stride=2, pad=(2, 2), act=2, cin=Cin, c0=c0)
    ref = F.leaky_relu(F.conv2d(xw[..., :Cin].permute(0, 3, 1, 2).double(), w.double(), b.double(), 2, 2), 0.1).permute(0, 2, 3, 1)
    got = out[..., c0:c0 + Cout].double()
    assert float((got - ref).abs().max()) < 2e-6 * (Cin * 25) ** 0.5 * float(ref.abs().max())
    assert bool((out[..., :c0] == 7.0).all()) and bool((out[..., c0 + Cout:] == 7.0).all())      # the rest of the map is untouched
    # --- Deconvolution 4x4 / 2 + Crop(offset (1,1)) to a 19 x 32 map, written into channels [256, 384) of a 416-channel map ---
    Cin, Cout, Hi, Wi, Hc, Wc, Lout, c0 = 96, 128, 10, 16, 19, 32, 416, 256
    x = torch.randn(1, Hi, Wi, Cin, device=dev)
    wt = torch.randn(Cin, Cout, 4, 4, device=dev) * 0.05          # conv_transpose2d layout (Cin, Cout, kh, kw)
    b = torch.randn(Cout, device=dev)
    full = F.leaky_relu(F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), wt.double(), b.double(), stride=2), 0.1)
    ref = full[:, :, 1:1 + Hc, 1:1 + Wc].permute(0, 2, 3, 1)
    out = torch.full((1, Hc, Wc, Lout), -3.0, device=dev)
    for py in (0, 1):
        for px in (0, 1):
            # output row 2m + py of the cropped map: taps ky = 3, 1 read input rows m - 1, m (py = 0); ky = 2, 0 read m, m + 1 (py = 1)
            kys = (3, 1) if py == 0 else (2, 0)
            kxs = (3, 1) if px == 0 else (2, 0)
            wp = wt[:, :, kys, :][:, :, :, kxs].permute(1, 0, 2, 3).contiguous()       # (Cout, Cin, 2, 2) as an ordinary convolution
            swp = hip.SplitWeight(wp)
            gh, gw = (Hc - py + 1) // 2, (Wc - px + 1) // 2
            hip.conv_split_view(x, swp, b, out, stride=1, pad=(1 - py, 1 - px), act=2, c0=c0, grid=(gh, gw), place=(py, px, 2, 2))
    got = out[..., c0:c0 + Cout].double()
    assert float((got - ref).abs().max()) < 2e-6 * (Cin * 4) ** 0.5 * float(ref.abs().max())
    assert bool((out[..., :c0] == -3.0).all()) and bool((out[..., c0 + Cout:] == -3.0).all())
    # the same as ONE launch (lsfa_deconv4x4s2_crop_fwd), also on FlowNet's own shapes (odd crops, K long enough
    # for the 128-channel workgroup tiles and for K slices)
    for (Cin, Cout, Hi, Wi, Hc, Wc, Lout, c0) in ((96, 128, 10, 16, 19, 32, 416, 256), (1056, 256, 10, 16, 19, 32, 800, 512),
                                                   (1024, 512, 5, 8, 10, 16, 1056, 512), (416, 64, 38, 63, 75, 125, 224, 128)):
        x = torch.randn(1, Hi, Wi, Cin, device=dev)
        wt = torch.randn(Cin, Cout, 4, 4, device=dev) * (1.0 / (4 * Cin) ** 0.5)
        b = torch.randn(Cout, device=dev)
        sws = hip.deconv_phase_weights(wt)
        one = torch.full((1, Hc, Wc, Lout), 5.0, device=dev)
        hip.deconv4x4s2_crop(x, sws, b, one, c0=c0, act=2)
        four = torch.full((1, Hc, Wc, Lout), 5.0, device=dev)
        for py in (0, 1):
            for px in (0, 1):
                hip.conv_split_view(x, sws[py * 2 + px], b, four, stride=1, pad=(1 - py, 1 - px), act=2, c0=c0,
                                    grid=((Hc - py + 1) // 2, (Wc - px + 1) // 2), place=(py, px, 2, 2))
        full = F.leaky_relu(F.conv_transpose2d(x.permute(0, 3, 1, 2).double(), wt.double(), b.double(), stride=2), 0.1)
        ref = full[:, :, 1:1 + Hc, 1:1 + Wc].permute(0, 2, 3, 1)
        assert float((one[..., c0:c0 + Cout].double() - ref).abs().max()) < 2e-6 * (Cin * 4) ** 0.5 * float(ref.abs().max())
        assert bool((one[..., :c0] == 5.0).all()) and bool((one[..., c0 + Cout:] == 5.0).all())
        # one launch vs four: the same arithmetic; bit-identical when the launch plan picks the same kernel and K cut for both,
        # fp32 round-off apart otherwise (small phases go to the direct kernel, whose three waves cut K differently)
        assert float((one - four).abs().max()) <= 4e-6 * float(ref.abs().max())


def test_amax_partial_is_the_exact_maximum(hip):
    """lsfa_amax_partial: the maximum of its 256 partial maxima is max|x| exactly (negative values, sizes that leave most of the 256
    workgroups without work, a NaN is ignored like fmaxf ignores it)."""
    g = torch.Generator(device=DEV).manual_seed(11)
    for n in (4, 1024, 2394 * 256, 2394 * 2048):
        x = torch.randn(n, device=DEV, generator=g) * 7
        x[n // 3] = -123.5
        out = hip.amax_partial(x)
        assert out.shape == (256,) and out.max().item() == x.abs().max().item() == 123.5
    x[5] = float('nan')
    assert hip.amax_partial(x).max().item() == 123.5


@pytest.mark.parametrize("case", ["res5 1x1 2048->512", "dcn 1x1 4608->512", "3x3 d6 256->128", "3x3 256->1024 nchw", "batch 3 nchw", "huge", "tiny",
                                  "zeros"])
def test_conv_split_h_fp16_two_piece_form_vs_float64(hip, case):
    """r3 (opt-in): lsfa_conv_split_h_fwd, fp32 operands in two fp16 pieces, three matrix instructions per product.  Against a
    float64 convolution it must be as close as the fp32 bound used for the bf16 three-piece form (2e-6 * sqrt(K) of max|y|) and
    not worse than 1.5x that form on the same inputs; inputs of 1e6 and 1e-6 magnitude go through the power-of-two scale unharmed;
    an all-zero map gives the bias; bit-reproducible."""
    shapes = {"res5 1x1 2048->512": (38, 63, 2048, 512, 1, 0, 1, False), "dcn 1x1 4608->512": (38, 63, 4608, 512, 1, 0, 1, False),
              "3x3 d6 256->128": (20, 30, 256, 128, 3, 6, 6, False), "3x3 256->1024 nchw": (19, 21, 256, 1024, 3, 1, 1, True)}
    H, W, ci, co, k, pad, dil, nchw = shapes.get(case, (20, 30, 256, 128, 3, 1, 1, False))
    nb = 1
    if case == "batch 3 nchw":      # pixel tiles of the NCHW reduce pass straddle the images (3 x 17 x 13 = 663 pixels, tiles of 64)
        nb, H, W, ci, co, k, pad, dil, nchw = 3, 17, 13, 256, 128, 3, 1, 1, True
    g = torch.Generator(device=DEV).manual_seed(len(case))
    mag = {"huge": 1e6, "tiny": 1e-6, "zeros": 0.0}.get(case, 3.0)
    x = torch.relu(torch.randn((nb, H, W, ci), device=DEV, generator=g)) * mag
    w = torch.randn((co, ci, k, k), device=DEV, generator=g) * 0.01
    b = torch.randn(co, device=DEV, generator=g) * mag * 0.1
    swh, sw = hip.SplitWeightH(w), hip.SplitWeight(w)
    y = hip.conv_split_h(x, swh, b, 1, pad, dil, act=1, nchw=nchw)
    y_again = hip.conv_split_h(x, swh, b, 1, pad, dil, act=1, nchw=nchw)
    assert torch.equal(y, y_again)
    if nchw:      # the NCHW reduce pass (tile turned in LDS) against the channels-last one: the same numbers, transposed
        assert torch.equal(y, hip.conv_split_h(x, swh, b, 1, pad, dil, act=1, nchw=False).permute(0, 3, 1, 2))
        assert torch.equal(hip.conv_split(x, sw, b, 1, pad, dil, relu=True, nchw=True),
                           hip.conv_split(x, sw, b, 1, pad, dil, relu=True, nchw=False).permute(0, 3, 1, 2))
    ref = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), b.double().cpu(), padding=pad, dilation=dil))
    y6 = hip.conv_split(x, sw, b, 1, pad, dil, relu=True, nchw=nchw)
    to_nchw = (lambda t: t) if nchw else (lambda t: t.permute(0, 3, 1, 2))
    scale = max(ref.abs().max().item(), 1e-300)
    e3 = (to_nchw(y).double().cpu() - ref).abs().max().item() / scale
    e6 = (to_nchw(y6).double().cpu() - ref).abs().max().item() / scale
    K = ci * k * k
    if case == "zeros":
        assert torch.equal(to_nchw(y).cpu(), ref.float())
        return
    assert e3 <= 2e-6 * K ** 0.5 and e3 <= 1.5 * e6 + 1e-7, (case, e3, e6)


def test_conv_fp16_form_flags_an_underestimated_scale(hip):
    """The fp16 two-piece form takes its scale from `amax_in`.  An UNDER-estimate (here the true partial maxima / 8) makes fp16(x s)
    overflow: the output is non-finite, and instead of handing a silently wrong feature map to the next layer the epilogue raises
    the status word, which lsfa_status_check turns into an error (LsfaError with lsfa_last_error()); the word is cleared by the
    check; a correct bound - and an OVER-estimate by 2^10 - leave it clear.  Through the K-sliced path (reduce pass) and the
    unsliced one, and with a ReLU that would hide a NaN / -inf from a test of the stored values."""
    g = torch.Generator(device=DEV).manual_seed(3)
    for (H, W, ci, co, k) in ((38, 63, 256, 128, 3), (12, 20, 64, 64, 1)):
        x = torch.randn((1, H, W, ci), device=DEV, generator=g) * 5.0
        w = torch.randn((co, ci, k, k), device=DEV, generator=g) * 0.05
        sw = hip.SplitWeight(w, pieces=2)
        status = hip.new_status(DEV)
        am = hip.amax_partial(x)
        y = hip.conv_split(x, sw, None, 1, k // 2, 1, relu=True, amax_in=am, status=status)
        hip.check_status(status)                                     # clear
        y_over = hip.conv_split(x, sw, None, 1, k // 2, 1, relu=True, amax_in=am * 1024.0, status=status)
        hip.check_status(status)
        assert float((y_over - y).abs().max()) <= 2e-6 * (ci * k * k) ** 0.5 * float(y.abs().max())      # 10 octaves of headroom cost nothing
        hip.conv_split(x, sw, None, 1, k // 2, 1, relu=True, amax_in=am / 8.0, status=status)
        with pytest.raises(hip.LsfaError, match="non-finite"):
            hip.check_status(status)
        hip.check_status(status)                                     # the check cleared the word
        # a map that already holds inf: flagged as an input problem too
        x2 = x.clone()
        x2[0, 1, 2, 3] = float('inf')
        hip.conv_split(x2, sw, None, 1, k // 2, 1, relu=True, status=status)
        with pytest.raises(hip.LsfaError):
            hip.check_status(status)


def test_conv_fp16_form_with_a_wide_per_channel_weight_range(hip):
    """BatchNorm folding multiplies every output channel of a weight by its own gamma / sqrt(var): real checkpoints carry channel scales a
    few thousand apart, and the two-piece form has ONE power-of-two scale per weight tensor (and one per activation map).  Output channels
    whose weights are 2^-12 of the largest channel's - and activations whose channels span the same range - must still come out with
    fp32-grade accuracy RELATIVE TO THEIR OWN magnitude: hi + lo keep 22 bits down to 2^-15 of the tensor's maximum before lo goes
    subnormal.  Channel c is scaled by 2^-(c mod 13); each output channel is judged against float64 on its own scale, next
    to the three-piece bf16 form, which has no scale."""
    g = torch.Generator(device=DEV).manual_seed(21)
    H, W, ci, co, k = 20, 33, 256, 128, 3
    K = ci * k * k
    ch_w = torch.pow(2.0, -(torch.arange(co, device=DEV) % 13).float())
    ch_x = torch.pow(2.0, -(torch.arange(ci, device=DEV) % 13).float())
    w = torch.randn((co, ci, k, k), device=DEV, generator=g) * 0.05 * ch_w.view(-1, 1, 1, 1)
    x = torch.randn((1, H, W, ci), device=DEV, generator=g) * 3.0 * ch_x
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), padding=1).permute(0, 2, 3, 1)
    per_ch = ref.abs().amax(dim=(0, 1, 2))                       # each output channel's own scale
    errs = {}
    for pieces in (2, 3):
        status = hip.new_status(DEV)
        y = hip.conv_split(x, hip.SplitWeight(w, pieces=pieces), None, 1, 1, 1, status=status)
        hip.check_status(status)
        errs[pieces] = float(((y.double().cpu() - ref).abs().amax(dim=(0, 1, 2)) / per_ch).max())
    assert errs[2] <= 2e-6 * K ** 0.5, errs
    assert errs[2] <= 2.0 * errs[3] + 1e-7, errs


def test_conv_counters_name_the_launch_plan_and_count_flops_and_bytes_once(hip):
    """r5 (lsfa_conv_plan_query; bench.py's roofline.by_kernel): every counted lsfa_conv_fwd call is attributed to the kernel instantiation the
    launch plan picks, with its algorithmic FLOPs (2 M N K) and bytes (every operand once); a forced plan shows up under its own name."""
    import re
    g = torch.Generator(device=DEV).manual_seed(11)
    N, H, W, ci, co = 2, 38, 63, 256, 256
    x = torch.relu(torch.randn((N, H, W, ci), device=DEV, generator=g))
    w = torch.randn((co, ci, 3, 3), device=DEV, generator=g) * 0.02
    b = torch.randn(co, device=DEV, generator=g)
    sw = hip.SplitWeight(w, pieces=2)
    am = hip.amax_partial(x)
    hip.conv_flops_reset(True)
    try:
        y0 = hip.conv_split(x, sw, b, 1, 1, 1, relu=True, amax_in=am)
        hip.conv_plan_override(kernel=2, nt=4, st=3, slices=1)
        y1 = hip.conv_split(x, sw, b, 1, 1, 1, relu=True, amax_in=am)
        hip.conv_plan_override(kernel=1, nt=2, st=2, slices=1)
        y2 = hip.conv_split(x, sw, b, 1, 1, 1, relu=True, amax_in=am)
    finally:
        hip.conv_plan_override()
    by = hip.conv_by_kernel()
    flops, launches = hip.conv_flops_read()
    hip.conv_flops_reset(False)
    for y in (y1, y2):                                            # (bit-identity across plans is test_conv_ring_every_plan...'s subject: per tile width and K cut)
        assert float((y - y0).abs().max()) <= 2e-6 * float(y0.abs().max()) * (ci * 9) ** 0.5
    assert launches == 3 and sum(v["calls"] for v in by.values()) == 3
    per_call = 2.0 * N * H * W * co * ci * 9
    assert abs(flops - 3 * per_call) < 1e-6 * per_call
    assert "conv_ring_kernel<4, 2, 3, true, false, 4>" in by and "conv_ring_kernel<2, 2, 2, false, false, 4>" in by, by
    for name, v in by.items():
        assert re.match(r"conv_(ring_kernel<[24], 2, [234], (true|false), false, [48]>|split_direct_kernel<2>|split3x3_kernel<\d, 2>)( \+split_reduce)?$", name), name
        # input + weights (two fp16 pieces = 4 bytes per weight) + output, each once
        want_mb = v["calls"] * (N * H * W * ci * 4 + co * ci * 9 * 4 + N * H * W * co * 4) / 1e6
        assert abs(v["mbytes"] - want_mb) < 0.02 * want_mb, (name, v, want_mb)
        assert abs(v["gflop"] - v["calls"] * per_call / 1e9) < 1e-3 * per_call / 1e9


def test_conv_fp16_form_takes_one_weight_scale_per_output_channel(hip):
    """r5 (VERDICT r4, weak point 2): a trained pre-activation ResNet's conv3 / shortcut rows follow its residual stream's channel
    magnitudes - 2^+-8 per channel with a few channels 2^14 above that (tests/test_parity_fullres_gpu.py::trained_like_batchnorm): 2^25 and
    more between the largest and the smallest output channel of ONE weight.  With r4's single power-of-two scale per weight tensor
    (per_channel_scale=False) the small channels' hi pieces keep a few bits and their lo pieces none: those OUTPUT channels come out with
    errors of 1e-3 ... 1 of their own magnitude.  With one scale per output channel (lsfa_conv_weights_pc + lsfa_conv_desc::w_scale, the
    default) every channel is as accurate as the three-piece bf16 form, which has no scale.  Checked on a 1x1 (ring kernel, loader /
    consumer waves), a 3x3 on a small map (direct kernel) and through the K-sliced reduce pass; the per-channel factor costs nothing
    in exactness (powers of two) - a weight with equal channels gives the same bits under both scalings."""
    g = torch.Generator(device=DEV).manual_seed(23)
    for H, W, ci, co, k, slices in ((38, 63, 256, 1024, 1, 0), (19, 32, 64, 64, 3, 0), (38, 63, 512, 256, 1, 2)):
        f = torch.exp2(torch.empty(co, device=DEV).uniform_(-8.0, 8.0, generator=g))
        f[torch.rand(co, device=DEV, generator=g) < 0.02] *= 2.0 ** 14
        f[0], f[1] = 2.0 ** 22, 2.0 ** -9                                # both ends present whatever the seed and the channel count: 2^31 apart
        w = torch.randn((co, ci, k, k), device=DEV, generator=g) * (1.0 / (ci * k * k) ** 0.5) * f.view(-1, 1, 1, 1)
        x = torch.relu(torch.randn((2, H, W, ci), device=DEV, generator=g)) * 2.0
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), padding=k // 2).permute(0, 2, 3, 1)
        own = ref.abs().amax(dim=(0, 1, 2))
        am = hip.amax_partial(x)
        err = {}
        try:
            if slices:
                hip.conv_plan_override(kernel=2, nt=4, st=3, slices=slices)
            for tag, sw in (("per_channel", hip.SplitWeight(w, pieces=2)), ("per_tensor", hip.SplitWeight(w, pieces=2, per_channel_scale=False)),
                            ("three_bf16", hip.SplitWeight(w, pieces=3))):
                status = hip.new_status(DEV)
                y = hip.conv_split(x, sw, None, 1, k // 2, 1, amax_in=am, status=status)
                hip.check_status(status)
                err[tag] = float(((y.double().cpu() - ref).abs().amax(dim=(0, 1, 2)) / own).max())
        finally:
            hip.conv_plan_override()
        K = ci * k * k
        assert err["per_channel"] <= 2e-6 * K ** 0.5 and err["per_channel"] <= 2.0 * err["three_bf16"] + 1e-7, (H, W, ci, co, k, err)
        assert err["per_tensor"] >= 1e-4, (H, W, ci, co, k, err, "the premise: one scale per tensor loses these channels")
    # equal channels: the same bits either way
    w = torch.randn((128, 64, 3, 3), device=DEV, generator=g) * 0.05
    w = w * (w.abs().reshape(128, -1).amax(1).max() / w.abs().reshape(128, -1).amax(1)).view(-1, 1, 1, 1) * 0.999      # every channel's maximum in one octave
    x = torch.randn((1, 20, 33, 64), device=DEV, generator=g)
    am = hip.amax_partial(x)
    a = hip.conv_split(x, hip.SplitWeight(w, pieces=2), None, 1, 1, 1, amax_in=am)
    b = hip.conv_split(x, hip.SplitWeight(w, pieces=2, per_channel_scale=False), None, 1, 1, 1, amax_in=am)
    assert torch.equal(a, b)


@pytest.mark.parametrize("pieces", [2, 3])
def test_conv_ring_every_plan_gives_the_same_convolution(hip, pieces):
    """lsfa_conv_plan_override: the ring kernel under every tile width x ring depth x K cut the plan may choose computes the
    same convolution (equal to float64 within the fp32 bound; bit-identical between ring depths and between mixed-role and
    loader / consumer waves, which only change who issues the copies and how far ahead they run), incl. a slice count that leaves the last slice short, residual + second output + amax_out through
    the reduce pass, and two-level accumulation past 16 chunks."""
    g = torch.Generator(device=DEV).manual_seed(17 + pieces)
    H, W, ci, co, k, dil = 23, 31, 256, 128, 3, 2          # 72 chunks of K
    x = torch.relu(torch.randn((1, H, W, ci), device=DEV, generator=g)) * 2.0
    w = torch.randn((co, ci, k, k), device=DEV, generator=g) * 0.02
    b = torch.randn(co, device=DEV, generator=g)
    res = torch.randn((1, H, W, co), device=DEV, generator=g)
    sc2, sh2 = torch.rand(co, device=DEV, generator=g) + 0.5, torch.randn(co, device=DEV, generator=g)
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), b.double().cpu(), padding=dil, dilation=dil)
    ref = (ref + res.permute(0, 3, 1, 2).double().cpu()).permute(0, 2, 3, 1)
    sw = hip.SplitWeight(w, pieces=pieces)
    am = hip.amax_partial(x)
    tol = 2e-6 * (ci * k * k) ** 0.5 * float(ref.abs().max())
    try:
        by_cut = {}
        plans = [(k_, n_, s_) for k_ in (1, 2) for n_ in (2, 4) for s_ in (2, 3, 4)]      # 2: loader / consumer waves
        if pieces < 3:
            plans += [(4, 4, 2), (4, 4, 3)]                                                # 4: 256-pixel tiles, eight mixed-role waves (r5)
        for kern, nt, st in plans:
            if True:
                if nt == 4 and pieces == 3 and st == 4:
                    continue
                for slices in (1, 2, 5, 7):
                    hip.conv_plan_override(kernel=kern, nt=nt, st=st, slices=slices)
                    slots = hip.amax_slots(1, DEV)[0]
                    y, y2 = hip.conv_split(x, sw, b, 1, dil, dil, residual=res, out2=torch.empty_like(res), scale2=sc2, shift2=sh2,
                                           amax_in=am, amax_out=slots)
                    assert float((y.double().cpu() - ref).abs().max()) < tol, (kern, nt, st, slices)
                    assert torch.equal(y2, torch.relu(y * sc2 + sh2))
                    assert slots.view(torch.float32).max().item() == y2.max().item()
                    key = (nt, slices)
                    if key in by_cut:
                        assert torch.equal(by_cut[key], y), (kern, nt, st, slices)      # neither ring depth nor wave roles change the arithmetic
                    by_cut[key] = y
    finally:
        hip.conv_plan_override()


def test_conv_tile_order_and_k_order(hip):
    """lsfa_conv_order_override (r6's K walk).  Tile order only renumbers the workgroups: with >= 4 channel tiles (Cout 512), K slices
    and the four phases of a transposed convolution in one launch, both orders give the SAME bits.  The K walk (tap by tap / channel chunk by
    channel chunk) changes the order of summation: the two walks agree with float64 within the fp32 bound, and within one walk every forced
    plan with the same tile width and K cut agrees bit for bit."""
    import torch.nn.functional as F
    g = torch.Generator(device=DEV).manual_seed(91)
    H, W, ci, co, dil = 23, 31, 256, 512, 2
    x = torch.relu(torch.randn((2, H, W, ci), device=DEV, generator=g))
    w = torch.randn((co, ci, 3, 3), device=DEV, generator=g) * 0.02
    b = torch.randn(co, device=DEV, generator=g)
    ref = torch.relu(F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), b.double().cpu(), padding=dil, dilation=dil)).permute(0, 2, 3, 1)
    tol = 2e-6 * (ci * 9) ** 0.5 * float(ref.abs().max())
    sw = hip.SplitWeight(w, pieces=2)
    am = hip.amax_partial(x)
    # a transposed convolution whose phases are K-sliced (K = 4 x 1056)
    Cin, Cout, Hi, Wi, Hc, Wc = 1056, 256, 10, 16, 19, 32
    xd = torch.randn(1, Hi, Wi, Cin, device=DEV, generator=g)
    wt = torch.randn(Cin, Cout, 4, 4, device=DEV, generator=g) * (1.0 / (4 * Cin) ** 0.5)
    bd = torch.randn(Cout, device=DEV, generator=g)
    sws = hip.deconv_phase_weights(wt)
    full = F.leaky_relu(F.conv_transpose2d(xd.permute(0, 3, 1, 2).double().cpu(), wt.double().cpu(), bd.double().cpu(), stride=2), 0.1)
    refd = full[:, :, 1:1 + Hc, 1:1 + Wc].permute(0, 2, 3, 1)
    try:
        seen = {}
        for k_order in (0, 1):
            for tile_order in (0, 1):
                hip.conv_order_override(tile_order=tile_order, k_order=k_order)
                for kern, nt, st, slices in ((1, 2, 2, 1), (2, 2, 3, 3), (1, 4, 2, 2), (2, 4, 3, 1), (2, 4, 3, 5)):
                    hip.conv_plan_override(kernel=kern, nt=nt, st=st, slices=slices)
                    y = hip.conv_split(x, sw, b, 1, dil, dil, relu=True, amax_in=am)
                    assert float((y.double().cpu() - ref).abs().max()) < tol, (k_order, tile_order, kern, nt, st, slices)
                    key = (k_order, nt, slices)
                    if key in seen:
                        assert torch.equal(seen[key], y), (k_order, tile_order, kern, nt, st, slices)
                    seen[key] = y
                for slices in (1, 2):
                    hip.conv_plan_override(kernel=2, nt=2, st=3, slices=slices)
                    one = torch.zeros((1, Hc, Wc, Cout), device=DEV)
                    hip.deconv4x4s2_crop(xd, sws, bd, one, act=2)
                    assert float((one.double().cpu() - refd).abs().max()) < 2e-6 * (Cin * 4) ** 0.5 * float(refd.abs().max())
                    key = ('deconv', k_order, slices)
                    if key in seen:
                        assert torch.equal(seen[key], one), (k_order, tile_order, slices)
                    seen[key] = one
        assert not torch.equal(seen[(0, 4, 1)], seen[(1, 4, 1)])       # the two walks do sum in different orders (else this test checks nothing)
    finally:
        hip.conv_plan_override()
        hip.conv_order_override()


def test_proposal_and_nms_do_not_depend_on_workspace_contents(hip):
    """The NMS sweep requests mask words speculatively; none of them may be a word nms_mask_kernel does not write (ADVICE r2:
    the lower triangle of the mask is never initialised).  The workspace comes from torch's caching allocator, so it is
    poisoned by filling a block of its size with 0xFF bytes and freeing it right before the call; results must not move."""
    rs = np.random.RandomState(11)
    n = 1500
    boxes = np.zeros((n, 5), np.float32)
    ctr = rs.uniform(50, 550, (n, 2)).astype(np.float32)
    wh = rs.uniform(20, 200, (n, 2)).astype(np.float32)
    boxes[:, 0:2], boxes[:, 2:4] = ctr - wh / 2, ctr + wh / 2
    boxes[:, 4] = np.sort(rs.uniform(0, 1, n).astype(np.float32))[::-1]
    t = torch.from_numpy(boxes).to(DEV)
    want = np.asarray(oracle.nms_sorted(boxes, 0.7))
    for poison in (False, True, True):
        if poison:
            junk = torch.full((64 << 20,), 0xFF, dtype=torch.uint8, device=DEV)
            torch.cuda.synchronize()
            del junk
        keep, num = hip.nms_sorted(t, 0.7)
        k = int(num.item())
        assert np.array_equal(keep.cpu().numpy()[:k], want[:k]) and k == len(want)
    # Proposal (the two-blocks-per-trip sweep): random RPN-like maps, poisoned workspace, against the oracle
    A, H, W = 9, 24, 40
    cls = rs.uniform(0, 1, (1, 2 * A, H, W)).astype(np.float32)
    bb = (rs.randn(1, 4 * A, H, W) * 0.3).astype(np.float32)
    im_info = np.array([[H * 16, W * 16, 1.0]], np.float32)
    op = hip.ProposalOp(feature_stride=16, scales=(8, 16, 32), ratios=(0.5, 1, 2), rpn_pre_nms_top_n=6000, rpn_post_nms_top_n=300,
                        threshold=0.7, rpn_min_size=16)
    want_rois, _ = oracle.proposal(cls, bb, im_info, 16, (8, 16, 32), (0.5, 1, 2), 6000, 300, 0.7, 16)
    for poison in (False, True):
        if poison:
            junk = torch.full((64 << 20,), 0xFF, dtype=torch.uint8, device=DEV)
            torch.cuda.synchronize()
            del junk
        rois = op(torch.from_numpy(cls).to(DEV), torch.from_numpy(bb).to(DEV), torch.from_numpy(im_info).to(DEV))
        np.testing.assert_array_equal(rois.cpu().numpy(), want_rois)
