"""Host-side preprocessing: product (torch) vs the numpy restatement."""
import numpy as np
import torch

from lsfa_amd.utils import image
from oracle import np_ref


def test_resize_scale_and_transform():
    rs = np.random.RandomState(0)
    im = rs.randint(0, 256, (72, 128, 3)).astype(np.float32)            # 1280x720 / 10
    out, scale = image.resize(torch.from_numpy(im), 60, 100, stride=0)
    assert scale == np_ref.resize_scale(im.shape, 60, 100) == 100.0 / 128
    ref = np_ref.cv2_resize_linear(im, scale, scale)
    assert out.shape == ref.shape == (56, 100, 3)
    np.testing.assert_allclose(out.numpy(), ref, rtol=2e-5, atol=5e-3)
    t = image.transform(out, [1.0, 2.0, 3.0], 0.5)
    np.testing.assert_allclose(t.numpy(), np_ref.transform(ref, [1.0, 2.0, 3.0], 0.5), rtol=2e-5, atol=5e-3)
    assert t.shape == (1, 3, 56, 100)
    padded, _ = image.resize(torch.from_numpy(im), 60, 100, stride=16)
    assert padded.shape == (64, 112, 3) and float(padded[60:].abs().sum()) == 0.0


def test_transform_mv_res_matches_restatement_including_channel_quirk():
    rs = np.random.RandomState(1)
    h, w = 72, 128
    mv = rs.randint(-20, 20, (h, w, 2)).astype(np.int32)
    res = rs.randint(-30, 30, (h, w, 3)).astype(np.int32)
    scale = 100.0 / 128
    for means, ps in (([0, 0, 0], 1.0), ([3.0, 5.0, 7.0], 0.5)):
        want_mv, want_res = np_ref.transform_mv_res(mv, res, scale, means, ps)
        got_mv, got_res = image.transform_mv_res(torch.from_numpy(mv), torch.from_numpy(res), scale, means, ps)
        assert got_mv.shape == want_mv.shape == (1, 2, 4, 7) and got_res.shape == (1, 3, 4, 7)
        # r5: the host path follows the reference's precisions (float32 first resize, float64 behind it, one rounding to float32): bit for bit
        np.testing.assert_array_equal(got_mv.numpy(), want_mv.astype(np.float32))
        np.testing.assert_array_equal(got_res.numpy(), want_res.astype(np.float32))
    # with zero means and unit scale the in-place loop leaves channel 2 == channel 0 (== source channel 2)
    _, r = image.transform_mv_res(torch.from_numpy(mv), torch.from_numpy(res), scale, [0, 0, 0], 1.0)
    np.testing.assert_array_equal(r[0, 0].numpy(), r[0, 2].numpy())


def test_resize_uses_the_given_factor_not_the_size_ratio():
    """cv2.resize(fx=, fy=) samples at (d + 0.5)/f - 0.5.  With f = 0.6 on 17 source pixels the output has
    cvRound(10.2) = 10 pixels and size ratio 1.7 != 1/f: a ramp image makes the two conventions differ by a
    known amount."""
    src = np.tile(np.arange(17, dtype=np.float32)[None, :, None], (3, 1, 1))          # value = x coordinate
    out = image._resize_hwc(torch.from_numpy(src), 0.6, 1.0)
    assert out.shape == (3, 10, 1)
    want = np.clip((np.arange(10) + 0.5) / 0.6 - 0.5, 0, 16)                            # bilinear of a ramp = the coordinate
    np.testing.assert_allclose(out[0, :, 0].numpy(), want, rtol=0, atol=1e-5)
    np.testing.assert_allclose(np_ref.cv2_resize_linear(src, 0.6, 1.0)[0, :, 0], want, rtol=0, atol=1e-5)
    assert abs(want[5] - ((5 + 0.5) * 1.7 - 0.5)) > 0.1                                  # the src/dst convention would differ
