"""BASELINE.json configs[2] and configs[4] at their real shapes (1000x600 frames, 38x63x1024 feature maps).

configs[2]  "bf16, key_interval=10, batch of 4 clips": four clips advance in lock-step, one image of each on
            the batch axis of every tensor (lsfa_amd.core.graphs, batch=4); dense contractions in bf16 (MFMA),
            hand-written stages fp32.  Every hand-written stage of every clip is pinned to the oracle bit for
            bit on the GPU's own inputs; the batched run agrees with the same clip run alone within bf16 round-off.
configs[4]  "key_interval=1 stress, HBM-bound warp roofline": (a) the warp and aggregate kernels on 32 maps per
            launch (942 MB, past the 256 MiB Infinity Cache) equal the oracle bit for bit; (b) a frame loop in which
            every frame is a key frame (flow warp x scale map + Nq aggregation per frame) through the stream pipeline.
"""
import numpy as np
import pytest
import torch

import oracle
from parity_util import check_dets, check_heads, check_key_frame, clone_dict, np_

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H, W = 600, 1000


@pytest.fixture(scope="module")
def world():
    from lsfa_amd.config.config import lsfa_test_config
    from lsfa_amd.symbols import params as P
    from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
    cfg = lsfa_test_config(key_frame_interval=10)
    arg, aux = P.init_params(cfg, seed=0)
    return dict(cfg=cfg, arg=arg, aux=aux, net=resnet_v1_101_flownet_rfcn(cfg))


def test_config3_bf16_four_clips_in_lockstep(world):
    from lsfa_amd.core.graphs import FrameGraphs
    from lsfa_amd.utils.synthetic import SyntheticClip
    cfg, arg, aux, net = world['cfg'], world['arg'], world['aux'], world['net']
    key = net.get_key_test_symbol(cfg).bind(arg, aux, DEV, torch.bfloat16)
    cur = net.get_cur_test_symbol(cfg).bind(arg, aux, DEV, torch.bfloat16)
    B = 4
    clips = [SyntheticClip(c, 4, H, W) for c in range(B)]
    cat = lambda fn: torch.cat([fn(c) for c in clips], 0)
    f0, f1, f2 = (cat(lambda c: c.frame(f, DEV)) for f in range(3))
    mv, res = cat(lambda c: c.motion_vector(2, 1, DEV)), cat(lambda c: c.res_diff(2, DEV))
    im_info = np.tile(clips[0].im_info(), (B, 1))
    R = cfg.TEST.RPN_POST_NMS_TOP_N
    fg = FrameGraphs(key, cur, cfg, H, W, DEV, use_graphs=False, prefetch=False, taps=True, batch=B)
    fg.first_frame(f0)
    fg.capture()
    feat0 = fg.feat.clone()
    assert feat0.shape == (B, 1024, 38, 63)
    dk, ck, _ = fg.key_frame(f1)
    ktaps, kout, kfeat = clone_dict(fg.key_taps), clone_dict(fg.key_out), fg.feat.clone()
    dk, ck = dk.cpu().numpy().copy(), ck.cpu().numpy().copy()
    dc, cc, _ = fg.cur_frame(f2, mv, res)
    ctaps, cout = clone_dict(fg.cur_taps), clone_dict(fg.cur_out)
    dc, cc = dc.cpu().numpy().copy(), cc.cpu().numpy().copy()
    assert dk.shape == (B, 31, R, 5) and ck.shape == (B, 31)
    # ---- key frame: flow warp x scale of every clip's own old feature, batched Nq combine, MultiProposal over 4 images
    warp_want = oracle.warp_bilinear(np_(feat0), np_(ktaps['flow']), mul=np_(ktaps['scale_map']))
    np.testing.assert_array_equal(np_(ktaps['warp']), warp_want)
    agg_want = oracle.aggregate_softmax2(warp_want, np_(ktaps['backbone_feat']), np_(ktaps['nq_logits']))
    np.testing.assert_array_equal(np_(kfeat), agg_want)
    check_heads(cfg, ktaps, kout, im_info)
    np.testing.assert_array_equal(np.unique(np_(kout['rois_output'])[:, 0]), np.arange(B))
    # ---- non-key frame
    want = oracle.warp_bilinear(np_(kfeat), np_(mv), add=np_(ctaps['small_feat']), res=np_(res),
                                res_w=arg['rnet_conv0_weight'], res_b=arg['rnet_conv0_bias'])
    np.testing.assert_array_equal(np_(cout['conv_feat']), want)
    check_heads(cfg, ctaps, cout, im_info)
    # ---- detections per clip
    for out, d, c in ((kout, dk, ck), (cout, dc, cc)):
        for b in range(B):
            sl = slice(b * R, (b + 1) * R)
            one = {'rois_output': out['rois_output'][sl], 'bbox_pred_reshape_output': out['bbox_pred_reshape_output'][:, sl],
                   'cls_prob_reshape_output': out['cls_prob_reshape_output'][:, sl]}
            check_dets(cfg, one, d[b], c[b], H, W)
    assert (np_(kfeat)[0] != np_(kfeat)[1]).any()                  # the clips do differ
    # ---- clip 2 alone (batch 1) through the same executors.  The launch plan (K slices) and with it the fp32 summation order may
    #      differ between batch 1 and batch 4; an fp32 difference in the last bits flips a bf16 rounding of the NEXT layer's operand
    #      now and then (0.4 % of that value) and ~100 layers compound the flips: measured 1.6 % of the feature's maximum.  Agreement is
    #      to bf16 round-off, not bit for bit
    solo = FrameGraphs(key, cur, cfg, H, W, DEV, use_graphs=False, prefetch=False, taps=False, batch=1)
    solo.first_frame(f0[2:3])
    solo.capture()
    solo.key_frame(f1[2:3])
    a, b = np_(solo.feat)[0], np_(kfeat)[2]
    assert np.abs(a - b).max() / np.abs(b).max() < 0.05
    # ---- the end-to-end criterion of the bf16 mode.  north_star words it as "mAP within 0.1 of the reference"; with the random
    #      weights of this test (no checkpoint in the image) RPN scores are near-ties by construction and ROI / mAP identity with an
    #      fp32 run is not a meaningful quantity (bench.py's parity block reports it: ~all 3000 ROIs differ).  What IS asserted:
    #      every hand-written stage bit-exact on the bf16 run's own inputs (above), no overflow flag, and the dense features of the
    #      first key frame within bf16 round-off of the fp32 oracle graph: the aggregated feature to 3 % of its maximum (one bf16
    #      rounding per operand is 2^-9 = 0.2 %; ~100 layers compound it), the RPN probabilities to 0.02 absolute.
    from oracle import graph_ref
    ref = graph_ref.key_forward(cfg, arg, aux, np_(f0[2:3]), np_(f0[2:3]), np.zeros((1, 1024, 1, 1), np.float32), clips[2].im_info())
    rel = np.abs(np_(feat0)[2] - ref['choose_feat_output'][0]).max() / np.abs(ref['choose_feat_output']).max()
    assert rel < 0.03, rel
    key.check_status()
    cur.check_status()


def test_config5_many_maps_per_launch_bit_exact():
    """32 feature maps per launch (> 256 MiB of traffic: the HBM-resident regime the roofline is quoted in)."""
    from lsfa_amd import hip
    M, C, h, w = 32, 1024, 38, 63
    g = torch.Generator(device=DEV).manual_seed(3)
    feat = torch.randn((M, C, h, w), device=DEV, generator=g)
    other = torch.randn((M, C, h, w), device=DEV, generator=g)
    flow = torch.randn((M, 2, h, w), device=DEV, generator=g) * 0.3 + torch.tensor([1.3, -0.8], device=DEV).view(1, 2, 1, 1)
    flow[5, :, 10:14] = 500.0                                     # one map with rows of flow that leave the map
    res = torch.randn((M, 3, h, w), device=DEV, generator=g)
    res_w, res_b = torch.randn((C, 3), device=DEV, generator=g) * 0.01, torch.randn(C, device=DEV, generator=g) * 0.01
    logits = torch.randn((2 * M, 1, h, w), device=DEV, generator=g) * 2
    got = hip.warp_bilinear(feat, flow, mul=other)
    np.testing.assert_array_equal(np_(got), oracle.warp_bilinear(np_(feat), np_(flow), mul=np_(other)))
    got = hip.warp_bilinear(feat, flow, add=other, res=res, res_w=res_w, res_b=res_b)
    np.testing.assert_array_equal(np_(got), oracle.warp_bilinear(np_(feat), np_(flow), add=np_(other), res=np_(res),
                                                                 res_w=np_(res_w), res_b=np_(res_b)))
    got = hip.aggregate_softmax2(feat, other, logits)
    np.testing.assert_array_equal(np_(got), oracle.aggregate_softmax2(np_(feat), np_(other), np_(logits)))


def test_config5_every_frame_a_key_frame(world):
    """key_interval = 1 through the stream pipeline at 1000x600: frames 1..4 are all key frames, each warping the
    previous frame's aggregated feature with FlowNet's flow (x scale map) and aggregating with Nq."""
    from lsfa_amd.core.graphs import FramePipeline
    from lsfa_amd.utils.synthetic import SyntheticClip
    cfg, arg, aux, net = world['cfg'], world['arg'], world['aux'], world['net']
    key = net.get_key_test_symbol(cfg).bind(arg, aux, DEV)
    cur = net.get_cur_test_symbol(cfg).bind(arg, aux, DEV)
    clip = SyntheticClip(7, 5, H, W, key_frame_interval=1)
    im_info = clip.im_info()
    frames = [clip.frame(f, DEV) for f in range(5)]
    outs = {}
    fp = FramePipeline(key, cur, cfg, H, W, DEV, lanes=1, taps=True)
    first = fp.first_frame(frames[0])
    outs[0] = dict(feat=fp.feat.clone())
    fp.capture()

    def keep(f):
        def deliver(bufs):
            lane = fp.delivering
            outs[f] = dict(taps=clone_dict(lane.taps), out=clone_dict(lane.out), feat=lane.feat.clone(),
                           dets=bufs[0].clone(), counts=bufs[1].clone())
        return deliver
    for f in range(1, 5):
        fp.key_frame(frames[f], deliver=keep(f))
    fp.join()
    torch.cuda.synchronize()
    for f in range(1, 5):
        check_key_frame(cfg, outs[f]['taps'], outs[f]['out'], outs[f - 1]['feat'], im_info)
        check_dets(cfg, outs[f]['out'], outs[f]['dets'].cpu().numpy(), outs[f]['counts'].cpu().numpy(), H, W)
