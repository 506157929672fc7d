"""The oracle against the golden vectors captured from the reference's own numpy
helpers (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

import oracle
from oracle import np_ref


def test_g1_anchor_table(golden):
    # C++ anchor generator (multi_proposal-inl.h:256-295) == numpy generator (generate_anchor.py)
    a = oracle.generate_anchors(16, (0.5, 1, 2), (8, 16, 32))
    assert a.dtype == np.float32
    np.testing.assert_array_equal(a.astype(np.float64), golden["g1_anchors"])
    np.testing.assert_array_equal(np_ref.generate_anchors(16, [0.5, 1, 2], np.array([8, 16, 32])), golden["g1_anchors"])
    np.testing.assert_array_equal(golden["g1_anchors"][0], [-84, -40, 99, 55])
    np.testing.assert_array_equal(golden["g1_anchors"][-1], [-168, -344, 183, 359])


@pytest.mark.parametrize("n", [50, 300, 2000])
@pytest.mark.parametrize("th", [0.3, 0.7])
def test_g2_nms_float64(golden, n, th):
    dets = golden["g2_n%d_float64_dets" % n]
    want = golden["g2_n%d_float64_keep_%02d" % (n, int(th * 10))]
    np.testing.assert_array_equal(np.asarray(np_ref.nms(dets, th)), want)
    np.testing.assert_array_equal(oracle.nms_f64(dets, th), want)


@pytest.mark.parametrize("n", [50, 300, 2000])
@pytest.mark.parametrize("th", [0.3, 0.7])
def test_g2_nms_float32_bitmask_equivalence(golden, n, th):
    """The CUDA bitmask NMS restatement (float32 devIoU, `> thresh` suppresses) returns the
    reference numpy nms's survivors on the same float32 boxes."""
    dets = golden["g2_n%d_float32_dets" % n]
    want = golden["g2_n%d_float32_keep_%02d" % (n, int(th * 10))]
    np.testing.assert_array_equal(np.asarray(np_ref.nms(dets, th)), want)
    np.testing.assert_array_equal(np.asarray(oracle.gpu_nms(dets, th)), want)
    # the materialised mask swept like nms_kernel.cu:133-146 agrees with the greedy loop
    order = np.argsort(-dets[:, 4], kind="stable")
    srt = dets[order]
    mask = oracle.nms_mask(srt, th)
    remv = np.zeros(mask.shape[1], np.uint64)
    keep = []
    for i in range(n):
        if not (int(remv[i // 64]) >> (i % 64)) & 1:
            keep.append(i)
            remv[i // 64:] |= mask[i, i // 64:]
    np.testing.assert_array_equal(order[keep], want)


def test_g2_empty(golden):
    assert golden["g2_empty_keep"].size == 0
    assert np_ref.nms(np.zeros((0, 5), np.float32), 0.3) == []
    assert oracle.nms_f64(np.zeros((0, 5)), 0.3).size == 0
    assert oracle.nms_sorted(np.zeros((0, 5), np.float32), 0.3).size == 0


def test_g3_bbox_pred_clip(golden):
    rois, deltas = golden["g3_rois"], golden["g3_deltas"]
    np.testing.assert_array_equal(np_ref.bbox_pred(rois, deltas), golden["g3_pred"])
    np.testing.assert_array_equal(np_ref.clip_boxes(np_ref.bbox_pred(rois, deltas), (600, 1000)), golden["g3_clip"])
    # C restatement: numpy's float32 exp is not correctly rounded, the oracle's is -> 1e-6 relative
    rois5 = np.hstack([np.zeros((rois.shape[0], 1), np.float32), rois])
    got = oracle.bbox_pred_clip(rois5, deltas, 600, 1000, 1.0)
    np.testing.assert_allclose(got, golden["g3_clip"], rtol=2e-6, atol=2e-4)
    assert golden["g3_empty"].shape == (0, 8)


def test_g4_iou_convention(golden):
    a, b = golden["g4_a"], golden["g4_b"]
    np.testing.assert_array_equal(np_ref.bbox_overlaps(a, b), golden["g4_iou"])
    # float32 devIoU agrees with the float64 matrix to float32 precision (+1 area convention)
    for i in range(0, 40, 7):
        for k in range(0, 25, 5):
            assert abs(oracle.dev_iou(a[i], b[k]) - golden["g4_iou"][i, k]) < 1e-5


def test_g5_frame_postprocess(golden):
    rois, deltas, probs = golden["g5_rois"], golden["g5_deltas"], golden["g5_probs"]
    scale = float(golden["g5_scale"])
    _, pred = np_ref.im_detect_post(rois, probs, deltas, (1, 3, 600, 1000), scale)
    np.testing.assert_array_equal(pred, golden["g5_pred_boxes"])
    all_boxes = np_ref.pred_eval_post(probs, pred, 31)
    np.testing.assert_array_equal([len(b) for b in all_boxes], golden["g5_counts"])
    np.testing.assert_array_equal(np.vstack(all_boxes), golden["g5_dets"])
    assert int(golden["g5_n_before_cap"]) > 300 and int(golden["g5_counts"].sum()) == 300
    # C restatement of the whole frame post-processing
    dets, counts, keep_idx = oracle.det_postprocess(rois, deltas, probs, 600, 1000, scale)
    np.testing.assert_array_equal(counts, golden["g5_counts"])
    got = np.vstack([dets[j, :counts[j]] for j in range(31)])
    np.testing.assert_allclose(got, golden["g5_dets"], rtol=2e-6, atol=2e-4)
    np.testing.assert_array_equal(got[:, 4], golden["g5_dets"][:, 4])


def test_key_frame_flags():
    # SURVEY.md A.5 state machine, dff_rfcn/core/loader.py:87-141
    f = np_ref.key_frame_flags([25], 10)
    assert f[0] == 0 and f[10] == 1 and f[20] == 1 and f[24] == 1
    assert all(x == 2 for i, x in enumerate(f) if i not in (0, 10, 20, 24))
    f = np_ref.key_frame_flags([144], 12)
    assert [i for i, x in enumerate(f) if x != 2] == list(range(0, 144, 12)) + [143]
    f = np_ref.key_frame_flags([3, 2], 10)
    assert f == [0, 2, 1, 0, 1]
    assert np_ref.shard_videos([10, 9, 8, 7, 1], 2) == [[0, 3, 4], [1, 2]]


def test_coviar_accumulation_hand_case():
    """coviar_data_loader.c:71-177 on a 12x10 frame: a 4x4 block moved by (+2,+1), then a later, overlapping 2x2
    block; second frame moves what the first one left.  Expected values worked out by hand from the C loops."""
    import oracle
    W, H = 12, 10
    accu = oracle.coviar_identity(W, H)
    assert accu[3, 7].tolist() == [7, 3]
    # {source, w, h, src_x, src_y, dst_x, dst_y}
    mvs = np.array([[-1, 4, 4, 4, 4, 6, 5],        # pixels dst x in [4,8), y in [3,7) take src x-2, y-1
                    [-1, 2, 2, 9, 2, 7, 4],        # later block: dst x in [6,8), y in [3,5) take src x+2, y-2
                    [-1, 4, 4, 5, 5, 5, 5],        # zero displacement: skipped (:92)
                    [-1, 4, 4, 1, 1, 0, 0]],       # partly outside: only in-bounds (dst AND src) pixels written
                   np.int32)
    a1 = oracle.coviar_accumulate(mvs, accu)
    assert a1[5, 5].tolist() == [3, 4]             # first block
    assert a1[3, 6].tolist() == [8, 1]             # overwritten by the second block (last writer wins)
    assert a1[6, 7].tolist() == [5, 5]             # first block, outside the second
    assert a1[8, 8].tolist() == [8, 8]             # untouched
    assert a1[0, 0].tolist() == [1, 1] and a1[1, 1].tolist() == [2, 2]     # block 4: dst (0,0) <- src (1,1)
    mv = oracle.coviar_mv(a1)
    assert mv[5, 5].tolist() == [2, 1] and mv[3, 6].tolist() == [-2, 2] and mv[8, 8].tolist() == [0, 0]
    # second frame reads the FIRST frame's result (accu_src_old), :111-113
    a2 = oracle.coviar_accumulate(np.array([[-1, 2, 2, 5, 5, 9, 8]], np.int32), a1)
    assert a2[8, 9].tolist() == a1[5, 5].tolist() == [3, 4]
    bgr0 = (np.arange(H * W * 3) % 251).astype(np.uint8).reshape(H, W, 3)
    bgr1 = ((np.arange(H * W * 3) * 7) % 253).astype(np.uint8).reshape(H, W, 3)
    res = oracle.coviar_residual(bgr1, bgr0, a2)
    assert res.dtype == np.int32
    assert res[8, 9].tolist() == (bgr1[8, 9].astype(np.int32) - bgr0[4, 3].astype(np.int32)).tolist()


def test_image_helpers_against_reference_g6(golden):
    """G6 (r5): lib/utils/image.py run here with cv2 / coviar_py2 stubbed (tests/golden/make_golden.py).  `transform` is the reference's
    own output; `transform_mv_res` / `resize` are the reference's code around the restated INTER_LINEAR, so everything but the
    interpolation arithmetic (padding to the stride, the in-place channel loop of :218-219, scales, transposes, the im_scale rule) is
    pinned.  The numpy restatement must reproduce them exactly, and so must the host path (lsfa_amd/utils/image.py) after the executor's rounding to
    float32 (`transform_mv_res`; the frame's own `resize` / `transform` stay float32 throughout: to round-off)."""
    import torch
    from lsfa_amd.utils import image
    im, means, ps = golden["g6_im"], golden["g6_means"], float(golden["g6_pixel_scale"])
    np.testing.assert_array_equal(np_ref.transform(im, means, ps), golden["g6_transform"])
    np.testing.assert_array_equal(np_ref.transform(im, np.zeros(3), 1.0), golden["g6_transform_zero_means"])
    imf = im.astype(np.float32) * np.float32(0.731)                   # a float32 frame, list means: float32 subtraction (config.py:177-182)
    np.testing.assert_array_equal(np_ref.transform(imf, [103.94, 116.78, 123.68], 0.017), golden["g6_transform_f32_list_means"])
    np.testing.assert_array_equal(np_ref.transform(imf, np.array([103.94, 116.78, 123.68]), 0.017), golden["g6_transform_f32_list_means"])
    np.testing.assert_array_equal(image.transform(torch.from_numpy(im), means, ps).numpy(), golden["g6_transform"].astype(np.float32))
    np.testing.assert_array_equal(image.transform(torch.from_numpy(imf), [103.94, 116.78, 123.68], 0.017).numpy(),
                                  golden["g6_transform_f32_list_means"].astype(np.float32))
    np.testing.assert_array_equal(image.transform(torch.from_numpy(im), np.zeros(3), 1.0).numpy(), golden["g6_transform_zero_means"].astype(np.float32))
    mv, res = golden["g6_mv"], golden["g6_res"]
    for tag, sc in (("s1", 1.0), ("s16", 1.6)):
        want_mv, want_res = golden["g6_mv_tensor_" + tag], golden["g6_res_tensor_" + tag]
        got_mv, got_res = np_ref.transform_mv_res(mv, res, sc, means, ps)
        np.testing.assert_array_equal(got_mv, want_mv)
        np.testing.assert_array_equal(got_res, want_res)
        h_mv, h_res = image.transform_mv_res(torch.from_numpy(mv), torch.from_numpy(res), sc, means, ps)
        assert tuple(h_mv.shape) == want_mv.shape and tuple(h_res.shape) == want_res.shape
        np.testing.assert_array_equal(h_mv.numpy(), want_mv.astype(np.float32))       # float64 arrays in the reference, float32 at the executor
        np.testing.assert_array_equal(h_res.numpy(), want_res.astype(np.float32))
    big = golden["g6_resize_in"]
    assert np_ref.resize_scale(big.shape, 60, 100) == float(golden["g6_resize_scale"]) == 2.0
    assert np_ref.resize_scale(big.shape, 60, 90) == float(golden["g6_resize_scale_capped"]) == 1.8
    out, sc = image.resize(torch.from_numpy(big), 60, 100, stride=16)
    assert sc == 2.0 and tuple(out.shape) == golden["g6_resize_out"].shape
    np.testing.assert_allclose(out.numpy(), golden["g6_resize_out"], rtol=1e-5, atol=1e-3)
    # a float32 frame padded to the stride is a float64 image: `transform` then subtracts the mean in float64; the host path follows
    padded, _ = image.resize(torch.from_numpy(big.astype(np.float32)), 60, 100, stride=16)
    assert padded.dtype == torch.float64
    np.testing.assert_array_equal(image.transform(padded, [103.94, 116.78, 123.68], 0.017).numpy(),
                                  golden["g6_resize_f32_stride16_transform"].astype(np.float32))
    out, sc = image.resize(torch.from_numpy(big), 60, 90)
    assert sc == 1.8 and tuple(out.shape) == golden["g6_resize_out_capped"].shape
    np.testing.assert_allclose(out.numpy(), golden["g6_resize_out_capped"], rtol=1e-5, atol=1e-3)


def test_cv2_uint8_fixed_point_restatement_invariants():
    """oracle/np_ref.py::cv2_resize_linear_u8 (OpenCV 3.2's fixed-point INTER_LINEAR for uint8 images, the reference's last-frame path; parity
    unpinned): the properties the published arithmetic guarantees - identity at scale 1, a constant image stays constant at any scale, and the
    result stays about one intensity level from the float interpolation of the same taps."""
    rs = np.random.RandomState(4)
    im = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    np.testing.assert_array_equal(np_ref.cv2_resize_linear_u8(im, 1.0, 1.0), im)
    const = np.full((20, 30, 3), 137, np.uint8)
    for f in (0.6, 0.78125, 1.25, 1.7):
        assert (np_ref.cv2_resize_linear_u8(const, f, f) == 137).all()
        a = np_ref.cv2_resize_linear_u8(im, f, f)
        b = np_ref.cv2_resize_linear(im.astype(np.float32), f, f)
        assert a.shape == b.shape and a.dtype == np.uint8
        assert np.abs(a.astype(np.float64) - b).max() <= 1.25      # final rounding 0.5 + two truncations 0.5 + 11-bit coefficients
    # the host path (lsfa_amd/utils/image.py) follows the same arithmetic: uint8 out, padded copy float64, float64 subtraction in transform
    import torch
    from lsfa_amd.utils import image
    big = rs.randint(0, 256, (30, 50, 3)).astype(np.uint8)
    out, sc = image.resize(torch.from_numpy(big), 60, 100, u8_fixed_point=True)
    assert sc == 2.0 and out.dtype == torch.uint8
    np.testing.assert_array_equal(out.numpy(), np_ref.cv2_resize_linear_u8(big, 2.0, 2.0))
    out2, sc2 = image.resize(torch.from_numpy(big), 60, 90, stride=16, u8_fixed_point=True)
    r = np_ref.cv2_resize_linear_u8(big, sc2, sc2)
    assert out2.dtype == torch.float64 and tuple(out2.shape[:2]) == (-(-r.shape[0] // 16) * 16, -(-r.shape[1] // 16) * 16)
    np.testing.assert_array_equal(out2.numpy()[:r.shape[0], :r.shape[1]], r.astype(np.float64))

