"""Key / cur graph parity on the GPU at a reduced resolution (192x320 -> 12x20 feature map).

Method (DESIGN.md "Parity method"): the dense contractions (MIOpen / hipBLASLt, BN folded)
are compared with the unfused torch-CPU statement within a tolerance; every hand-written
stage is then checked BIT-EXACTLY by feeding the oracle the GPU's own inputs to that stage
("teacher forcing"), so a 1-ulp difference in a convolution cannot hide or fake a kernel bug.
"""
import gc

import numpy as np
import pytest
import torch

import oracle
from oracle import graph_ref

pytestmark = pytest.mark.gpu
# dense stages (MFMA contractions with BN folded in float64) against the unfused torch-CPU fp32 graph, relative to the
# map's maximum: both sides are fp32 with different summation orders; measured 1.4e-6 ... 2.6e-6
TOL_DENSE = 2e-5
DEV = "cuda:0"
H, W = 192, 320


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


@pytest.fixture(scope="module")
def world():
    from lsfa_amd.config.config import lsfa_test_config
    from lsfa_amd.symbols import params as P
    from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
    from lsfa_amd.utils.synthetic import SyntheticClip
    cfg = lsfa_test_config(key_frame_interval=10)
    arg, aux = P.init_params(cfg, seed=3)
    net = resnet_v1_101_flownet_rfcn(cfg)
    key = net.get_key_test_symbol(cfg).bind(arg, aux, DEV)
    cur = net.get_cur_test_symbol(cfg).bind(arg, aux, DEV)
    clip = SyntheticClip(0, 12, H, W)
    return dict(cfg=cfg, arg=arg, aux=aux, key=key, cur=cur, clip=clip)


def np_(t):
    return t.detach().float().cpu().numpy()


def check_heads(cfg, taps, out, im_info):
    rois, _ = oracle.proposal(np_(taps['rpn_cls_prob']), np_(taps['rpn_bbox_pred']), im_info, 16, cfg.network.ANCHOR_SCALES,
                              cfg.network.ANCHOR_RATIOS, cfg.TEST.RPN_PRE_NMS_TOP_N, cfg.TEST.RPN_POST_NMS_TOP_N,
                              cfg.TEST.RPN_NMS_THRESH, cfg.TEST.RPN_MIN_SIZE)
    np.testing.assert_array_equal(np_(out['rois_output']), rois)
    cls_prob, _, bbox_pred = oracle.rfcn_head(np_(taps['cls_map']), np_(taps['box_map']), rois)
    np.testing.assert_array_equal(np_(out['cls_prob_reshape_output'])[0], cls_prob)
    np.testing.assert_array_equal(np_(out['bbox_pred_reshape_output'])[0], bbox_pred)


def test_key_and_cur_graphs(world):
    cfg, arg, aux, key, cur, clip = (world[k] for k in ('cfg', 'arg', 'aux', 'key', 'cur', 'clip'))
    im_info = clip.im_info()
    im_info_t = torch.from_numpy(im_info).to(DEV)
    f0, f10 = clip.frame(0), clip.frame(10)
    placeholder = torch.zeros(1, 1024, 1, 1, device=DEV)

    # ---- first frame (flag 0): backbone + heads, no aggregation -----------------------
    key.taps = {}
    out0 = key.forward(data=f0.to(DEV), im_info=im_info_t, data_key_old=f0.to(DEV), feat_key_old=placeholder)
    ref0 = graph_ref.key_forward(cfg, arg, aux, f0.numpy(), f0.numpy(), np.zeros((1, 1024, 1, 1), np.float32), im_info)
    assert out0['choose_feat_output'].shape == (1, 1024, 12, 20)
    assert rel_err(np_(key.taps['backbone_feat']), ref0['backbone_feat']) < TOL_DENSE
    assert rel_err(np_(key.taps['cls_map']), ref0['cls_map']) < TOL_DENSE
    assert rel_err(np_(key.taps['rpn_bbox_pred']), ref0['rpn_bbox_pred']) < TOL_DENSE
    assert np.abs(np_(key.taps['rpn_cls_prob']) - ref0['rpn_cls_prob']).max() < 1e-4
    check_heads(cfg, key.taps, out0, im_info)
    feat0 = out0['choose_feat_output']

    # ---- non-key frame (flag 2): MV warp + residual + small net ------------------------
    f3 = clip.frame(3)
    mv, res = clip.motion_vector(3, 0), clip.res_diff(3)
    cur.taps = {}
    out3 = cur.forward(data=f3.to(DEV), im_info=im_info_t, feat_key=feat0, motion_vector=mv.to(DEV), res_diff=res.to(DEV))
    ref_small = graph_ref.small_net_feature(graph_ref.Params(arg, aux), f3).numpy()
    assert rel_err(np_(cur.taps['small_feat']), ref_small) < TOL_DENSE
    want = oracle.warp_bilinear(np_(feat0), mv.numpy(), add=np_(cur.taps['small_feat']), res=res.numpy(),
                                res_w=arg['rnet_conv0_weight'], res_b=arg['rnet_conv0_bias'])
    np.testing.assert_array_equal(np_(out3['conv_feat']), want)
    check_heads(cfg, cur.taps, out3, im_info)

    # ---- second key frame (flag 1): FlowNet + flow warp x scale + Nq aggregation --------
    key.taps = {}
    out10 = key.forward(data=f10.to(DEV), im_info=im_info_t, data_key_old=f0.to(DEV), feat_key_old=feat0)
    p = graph_ref.Params(arg, aux)
    flow_ref, scale_ref = graph_ref.get_flownet(p, f10, f0)
    assert np.abs(np_(key.taps['flow']) - flow_ref.numpy()).max() < TOL_DENSE * max(1.0, float(flow_ref.abs().max()))
    assert rel_err(np_(key.taps['scale_map']), scale_ref.numpy()) < TOL_DENSE
    warp_want = oracle.warp_bilinear(np_(feat0), np_(key.taps['flow']), mul=np_(key.taps['scale_map']))
    np.testing.assert_array_equal(np_(key.taps['warp']), warp_want)
    logits_ref = graph_ref.nq_logits(p, torch.from_numpy(warp_want), key.taps['backbone_feat'].cpu()).numpy()
    assert np.abs(np_(key.taps['nq_logits']) - logits_ref).max() < TOL_DENSE * max(1.0, float(np.abs(logits_ref).max()))
    agg_want = oracle.aggregate_softmax2(warp_want, np_(key.taps['backbone_feat']), np_(key.taps['nq_logits']))
    np.testing.assert_array_equal(np_(out10['choose_feat_output']), agg_want)
    check_heads(cfg, key.taps, out10, im_info)


def test_pred_eval_loop_flags_and_shapes(world):
    """The frame loop (TestLoader + pred_eval) on a 12-frame clip with interval 5."""
    from lsfa_amd.config.config import lsfa_test_config
    from lsfa_amd.core.loader import TestLoader
    from lsfa_amd.core.tester import Predictor, pred_eval
    from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
    from lsfa_amd.utils.synthetic import synthetic_roidb
    cfg = lsfa_test_config(key_frame_interval=5)
    roidb = synthetic_roidb(1, 12, H, W, 5)
    loader = TestLoader(roidb, cfg, device=DEV)
    flags = []
    probe = TestLoader(roidb, cfg, device=DEV)
    for _, flag, _ in probe:
        flags.append(flag)
    assert flags == oracle_flags([12], 5)
    net = resnet_v1_101_flownet_rfcn(cfg)
    names = loader.data_name
    kp = Predictor(net.get_key_test_symbol(cfg), names, None, DEV, arg_params=world['arg'], aux_params=world['aux'])
    cp = Predictor(net.get_cur_test_symbol(cfg), names, None, DEV, arg_params=world['arg'], aux_params=world['aux'])
    all_boxes, frame_ids = pred_eval(0, kp, cp, loader, None, cfg)
    assert list(frame_ids) == list(range(12))
    assert len(all_boxes) == 31 and len(all_boxes[1]) == 12
    per_frame = [sum(len(all_boxes[j][i]) for j in range(1, 31)) for i in range(12)]
    assert all(0 < n <= 300 + 30 for n in per_frame)
    b = np.vstack([all_boxes[j][3] for j in range(1, 31)])
    assert (b[:, 0] >= 0).all() and (b[:, 2] <= W - 1).all() and (b[:, 3] <= H - 1).all() and (b[:, 4] > 1e-4).all()


def oracle_flags(seg_lens, k):
    from oracle import np_ref
    return np_ref.key_frame_flags(seg_lens, k)


def test_hipgraph_replay_equals_eager(world):
    """The captured per-frame hipGraphs, with and without the small-net prefetch fork, reproduce the eager
    launch sequence BIT FOR BIT (every kernel is this library's own: nothing chooses an algorithm by how a call is issued), and every
    frame of the eager run is pinned to the oracle stage by stage.  The schedule starts 0, 2, 2, 1: a
    non-key frame right after the first frame — with prefetch on, its small-net feature must be that
    frame's, not a left-over of capture's warm-up."""
    from parity_util import assert_dets_equal, check_cur_frame, check_dets, check_key_frame, clone_dict
    from lsfa_amd.core.graphs import FrameGraphs
    cfg, arg, key, cur, clip = world['cfg'], world['arg'], world['key'], world['cur'], world['clip']
    key.taps = cur.taps = None
    im_info = clip.im_info()
    # (frame, its key frame): cur, cur, key, cur, cur, key, cur, key
    sched = ((1, 0), (2, 0), (3, 3), (4, 3), (5, 3), (6, 6), (7, 6), (8, 8))
    results = []
    for use_graphs, prefetch, first_next in ((False, False, False), (True, False, False), (True, True, True), (True, True, False)):
        fg = FrameGraphs(key, cur, cfg, H, W, DEV, use_graphs=use_graphs, prefetch=prefetch, taps=True)
        fg.first_frame(clip.frame(0, DEV), clip.frame(1, DEV) if first_next else None)
        fg.capture()
        out = []
        for idx, (f, kf) in enumerate(sched):
            nxt = clip.frame(f + 1, DEV) if idx + 1 < len(sched) and sched[idx + 1][0] != sched[idx + 1][1] else None
            if f == kf:
                prev_feat = fg.feat_old.clone()
                d, c, k = fg.key_frame(clip.frame(f, DEV), nxt)
                rec = dict(kind='key', prev_feat=prev_feat, taps=clone_dict(fg.key_taps), out=clone_dict(fg.key_out))
            else:
                mv, res = clip.motion_vector(f, kf, DEV), clip.res_diff(f, DEV)
                d, c, k = fg.cur_frame(clip.frame(f, DEV), mv, res, nxt)
                rec = dict(kind='cur', mv=mv, res=res, taps=clone_dict(fg.cur_taps), out=clone_dict(fg.cur_out))
            rec.update(dets=d.cpu().numpy().copy(), counts=c.cpu().numpy().copy(), feat=fg.feat.clone())
            out.append(rec)
        results.append(out)
    # the eager run, frame by frame, against the oracle
    for (f, kf), r in zip(sched, results[0]):
        if r['kind'] == 'key':
            check_key_frame(cfg, r['taps'], r['out'], r['prev_feat'], im_info)
        else:
            check_cur_frame(cfg, arg, r['taps'], r['out'], r['feat'], r['mv'], r['res'], im_info)
        check_dets(cfg, r['out'], r['dets'], r['counts'], H, W)
    # graph replay == eager, exactly
    for v in range(1, len(results)):
        for (f, kf), r0, r1 in zip(sched, results[0], results[v]):
            tag = "variant %d, frame %d" % (v, f)
            for name in r0['taps']:
                assert torch.equal(r0['taps'][name], r1['taps'][name]), "%s: tap %s differs" % (tag, name)
            for name in r0['out']:
                assert torch.equal(r0['out'][name], r1['out'][name]), "%s: output %s differs" % (tag, name)
            assert_dets_equal(r0['dets'], r0['counts'], r1['dets'], r1['counts'], tag)


@pytest.mark.parametrize("pipeline", [False, True])
def test_clip_end_to_end_map_vs_oracle(world, pipeline):
    """Whole path on a 7-frame 192x320 clip, interval 3: GPU (test_rcnn -> pred_eval, and the
    stream-pipelined pred_eval_pipelined) vs the CPU
    oracle's frame loop.  The convolutions differ by fp32 round-off, so individual low-score
    detections may swap; the agreement metric is VID mAP@0.5 of the GPU detections against the
    oracle's confident detections taken as ground truth: must be within 0.1 of 1."""
    from oracle import frame_loop
    from lsfa_amd.config.config import lsfa_test_config
    from lsfa_amd.dataset import vid_eval as ve
    from lsfa_amd.function.test_rcnn import test_rcnn
    from lsfa_amd.utils.synthetic import synthetic_roidb
    cfg = lsfa_test_config(key_frame_interval=3)
    arg, aux = world['arg'], world['aux']
    roidb = synthetic_roidb(1, 7, H, W, 3)
    rows_gpu, frame_ids = test_rcnn(cfg, roidb, arg, aux, device=DEV, pipeline=pipeline)
    clip = roidb[0]['clip']
    rows_ref = frame_loop.run_clip(cfg, arg, aux, 7, lambda f: clip.frame(f).numpy(),
                                   lambda f, k: clip.motion_vector(f, k).numpy(), lambda f: clip.res_diff(f).numpy(),
                                   clip.im_info())
    assert list(frame_ids) == list(range(7))
    assert abs(len(rows_gpu) - len(rows_ref)) <= 0.05 * len(rows_ref)
    # ground truth := the oracle's 5 most confident detections per frame
    gt = []
    for f in range(7):
        r = rows_ref[rows_ref[:, 0] == f]
        r = r[np.argsort(-r[:, 2], kind='stable')][:5]
        gt.append({'img_id': f, 'bbox': r[:, 3:7], 'label': r[:, 1].astype(int)})
    ap_ref = ve.vid_eval(rows_ref, gt, 31)
    ap_gpu = ve.vid_eval(rows_gpu, gt, 31)
    present = np.unique(np.concatenate([g['label'] for g in gt])) - 1
    assert abs(ap_gpu[present].mean() - ap_ref[present].mean()) < 0.1
    assert ap_ref[present].mean() > 0.5


def test_bf16_contractions_config3(world):
    """BASELINE configs[2]: bf16 MFMA for the dense contractions, fp32 hand-written kernels.  The
    feature map stays within bf16 round-off of the fp32 run and every hand-written stage is still
    bit-exact on the inputs it was given."""
    from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
    cfg, arg, aux, clip = world['cfg'], world['arg'], world['aux'], world['clip']
    net = resnet_v1_101_flownet_rfcn(cfg)
    key16 = net.get_key_test_symbol(cfg).bind(arg, aux, DEV, dtype=torch.bfloat16)
    cur16 = net.get_cur_test_symbol(cfg).bind(arg, aux, DEV, dtype=torch.bfloat16)
    im_info = clip.im_info()
    im_t = torch.from_numpy(im_info).to(DEV)
    f0 = clip.frame(0, DEV)
    ph = torch.zeros(1, 1024, 1, 1, device=DEV)
    key16.taps = {}
    o16 = key16.forward(data=f0, im_info=im_t, data_key_old=f0, feat_key_old=ph)
    world['key'].taps = None
    o32 = world['key'].forward(data=f0, im_info=im_t, data_key_old=f0, feat_key_old=ph)
    a, b = np_(o16['choose_feat_output']), np_(o32['choose_feat_output'])
    assert o16['choose_feat_output'].dtype == torch.float32
    assert np.abs(a - b).max() / np.abs(b).max() < 0.08
    check_heads(cfg, key16.taps, o16, im_info)
    cur16.taps = {}
    mv, res = clip.motion_vector(2, 0, DEV), clip.res_diff(2, DEV)
    o2 = cur16.forward(data=clip.frame(2, DEV), im_info=im_t, feat_key=o16['choose_feat_output'], motion_vector=mv, res_diff=res)
    want = oracle.warp_bilinear(a, np_(mv), add=np_(cur16.taps['small_feat']), res=np_(res), res_w=arg['rnet_conv0_weight'],
                                res_b=arg['rnet_conv0_bias'])
    np.testing.assert_array_equal(np_(o2['conv_feat']), want)
    check_heads(cfg, cur16.taps, o2, im_info)
    key16.taps = {}
    o10 = key16.forward(data=clip.frame(10, DEV), im_info=im_t, data_key_old=f0, feat_key_old=o16['choose_feat_output'])
    assert torch.isfinite(o10['choose_feat_output']).all()
    check_heads(cfg, key16.taps, o10, im_info)


def test_batch_symbol_tile_as_multiproposal(world):
    """get_batch_test_symbol: 1 key + 2 other frames in one pass (DFF-style batch mode): MultiProposal over
    3 images, PSROI with per-roi batch indices, im_batch_detect."""
    from lsfa_amd.core.loader import DataBatch
    from lsfa_amd.core.tester import Predictor, im_batch_detect
    from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
    cfg, arg, aux, clip = world['cfg'], world['arg'], world['aux'], world['clip']
    net = resnet_v1_101_flownet_rfcn(cfg)
    sym = net.get_batch_test_symbol(cfg)
    names = sym.data_names
    pred = Predictor(sym, names, None, DEV, arg_params=arg, aux_params=aux)
    data_key = clip.frame(0, DEV)
    data_other = torch.cat([clip.frame(1, DEV), clip.frame(2, DEV)], 0)
    im_info = np.tile(clip.im_info(), (3, 1)).astype(np.float32)
    im_info[2, 2] = 1.0
    batch = DataBatch(data=[[data_key, data_other, torch.from_numpy(im_info).to(DEV)]])
    pred._exec.taps = {}
    out = pred.predict(batch)[0]
    taps = pred._exec.taps
    assert out['rois_output'].shape == (900, 5) and out['cls_prob_reshape_output'].shape == (1, 900, 31)
    ref = graph_ref.batch_forward(cfg, arg, aux, data_key.cpu().numpy(), data_other.cpu().numpy(), im_info)
    assert rel_err(np_(taps['backbone_feat']), ref['backbone_feat']) < TOL_DENSE
    assert np.abs(np_(taps['flow']) - ref['flow']).max() < TOL_DENSE * max(1.0, np.abs(ref['flow']).max())
    want_warp = oracle.warp_bilinear(np_(taps['backbone_feat']), np_(taps['flow']), mul=np_(taps['scale_map']))
    np.testing.assert_array_equal(np_(taps['warp']), want_warp)                 # feat_n = 1 broadcast == tile_as
    check_heads(cfg, taps, out, im_info)                                         # MultiProposal + PSROI over 3 images
    np.testing.assert_array_equal(np.unique(np_(out['rois_output'])[:, 0]), [0, 1, 2])
    scores_all, boxes_all, _ = im_batch_detect(pred, batch, names, [im_info[:, 2]], cfg)
    assert len(scores_all) == 3 and boxes_all[1].shape == (300, 8) and boxes_all[1].dtype == np.float64
    want = oracle.bbox_pred_clip(np_(out['rois_output'])[300:600], np_(out['bbox_pred_reshape_output'])[0, 300:600], H, W, 1.0)
    # im_batch_detect runs its own forward; MIOpen may pick another algorithm on the 2nd call -> fp32 round-off
    np.testing.assert_allclose(boxes_all[1], want, rtol=1e-5, atol=2e-3)


@pytest.mark.parametrize("mode", ["fgfa", "average"])
def test_key_graph_other_aggregations(mode):
    """get_key_test_symbol's other two aggregation branches (symbols/resnet_v1_101_flownet_rfcn.py:312-315):
    Fgfa cosine-similarity weights (:132-148) and the plain 0.5*(warp + cur) mean."""
    from lsfa_amd.config.config import lsfa_test_config
    from lsfa_amd.symbols import params as P
    from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
    from lsfa_amd.utils.synthetic import SyntheticClip
    cfg = lsfa_test_config(key_frame_interval=10)
    cfg.network.add_Nq_net = False
    cfg.network.add_Fgfa_net = (mode == "fgfa")
    arg, aux = P.init_params(cfg, seed=5)
    key = resnet_v1_101_flownet_rfcn(cfg).get_key_test_symbol(cfg).bind(arg, aux, DEV)
    clip = SyntheticClip(1, 12, H, W)
    im_info = clip.im_info()
    im_info_t = torch.from_numpy(im_info).to(DEV)
    f0, f10 = clip.frame(0).to(DEV), clip.frame(10).to(DEV)
    feat0 = key.forward(data=f0, im_info=im_info_t, data_key_old=f0,
                        feat_key_old=torch.zeros(1, 1024, 1, 1, device=DEV))['choose_feat_output']
    key.taps = {}
    out = key.forward(data=f10, im_info=im_info_t, data_key_old=f0, feat_key_old=feat0)
    warp = np_(key.taps['warp'])
    cur_feat = np_(key.taps['backbone_feat'])
    if mode == "fgfa":
        p = graph_ref.Params(arg, aux)
        emb_ref = graph_ref.embed(p, torch.from_numpy(cur_feat), torch.from_numpy(warp)).numpy()
        emb = np_(key.taps['embed'])
        assert emb.shape == (2, 2048, 12, 20)
        assert rel_err(emb, emb_ref) < TOL_DENSE
        want = oracle.aggregate_cosine(warp, cur_feat, emb[1:2], emb[0:1])
        np.testing.assert_array_equal(np_(out['choose_feat_output']), want)
    else:
        want = (0.5 * (torch.from_numpy(warp) + torch.from_numpy(cur_feat))).numpy()
        np.testing.assert_array_equal(np_(out['choose_feat_output']), want)
    check_heads(cfg, key.taps, out, im_info)
    # whole-graph statement with the same switches
    ref = graph_ref.key_forward(cfg, arg, aux, np_(f10), np_(f0), np_(feat0), im_info)
    assert rel_err(np_(out['choose_feat_output']), ref['choose_feat_output']) < TOL_DENSE


def test_demo_frame_directory(tmp_path, monkeypatch):
    """lsfa_amd.demo over a directory of PNG frames + per-frame mv/res files (demo.py:63-158 loop):
    frame 0 eager, key frames every `interval`, non-key frames in between, hipGraph replay ==
    eager launch of the same loop."""
    import json
    import sys
    from PIL import Image
    from lsfa_amd import demo
    rs = np.random.RandomState(0)
    fdir, mdir = tmp_path / "frames", tmp_path / "mv"
    fdir.mkdir(); mdir.mkdir()
    base = rs.randint(0, 255, (120, 200, 3)).astype(np.uint8)
    for i in range(7):
        Image.fromarray(np.roll(base, 3 * i, axis=1)).save(str(fdir / ("%06d.png" % i)))
        np.savez(str(mdir / ("%06d.npz" % i)), mv=rs.uniform(-3, 3, (120, 200, 2)).astype(np.float32),
                 res=rs.uniform(-20, 20, (120, 200, 3)).astype(np.float32))
    outs = []
    for extra in ([], ["--no-graph"]):
        out = tmp_path / ("dets%d.json" % len(outs))
        monkeypatch.setattr(sys, "argv", ["demo", "--frames", str(fdir), "--mv", str(mdir), "--interval", "3",
                                          "--score", "0.05", "--out", str(out)] + extra)
        demo.main()
        outs.append(json.loads(out.read_text()))
    a, b = outs
    assert [r["key"] for r in a] == [True, False, False, True, False, False, True]
    assert len(a) == len(b) == 7
    n_a, n_b = sum(len(r["dets"]) for r in a), sum(len(r["dets"]) for r in b)
    assert n_a > 0 and abs(n_a - n_b) <= 0.3 * max(n_a, n_b)
    # frame 0 runs eagerly in both: same detections up to the convolution library's algorithm choice
    da, db = a[0]["dets"], b[0]["dets"]
    assert len(da) == len(db)
    top_a, top_b = max(da, key=lambda d: d["score"]), max(db, key=lambda d: d["score"])
    assert top_a["class"] == top_b["class"] and abs(top_a["score"] - top_b["score"]) < 1e-4
    np.testing.assert_allclose(top_a["box"], top_b["box"], atol=1e-2)


def _run_clip_through_pipeline(fp, frames, sched, mvs, ress):
    """Push frame 0 + `sched` through a FramePipeline(taps=True); every frame's taps, outputs, detections and
    the key feature it produced are cloned on the frame's own stream inside `deliver`."""
    from parity_util import clone_dict
    outs = {}

    def keep(f, is_key):
        def deliver(bufs):
            lane = fp.delivering
            if is_key:
                o = dict(taps=clone_dict(lane.taps), out=clone_dict(lane.out), feat=lane.feat.clone())
            else:
                o = dict(taps=clone_dict(lane.cur_taps), out=clone_dict(lane.cur_out))
            o.update(dets=bufs[0].clone(), counts=bufs[1].clone())
            outs[f] = o
        return deliver
    first = fp.first_frame(frames[0])
    outs[0] = dict(dets=first[0].clone(), counts=first[1].clone(), feat=fp.feat.clone())
    if not fp.captured:
        fp.capture()
    for f, kf in sched:
        if f == kf:
            fp.key_frame(frames[f], deliver=keep(f, True))
        else:
            fp.cur_frame(frames[f], mvs[f], ress[f], deliver=keep(f, False))
    fp.join()
    torch.cuda.synchronize()
    return outs


@pytest.mark.parametrize("lookahead,layout", [(False, "probe"), (True, "probe"), (False, "plain"), (False, "one-queue"), (True, "one-queue")])
def test_frame_pipeline_frames_pinned_to_oracle_and_to_the_serial_loop(world, lookahead, layout):
    """FramePipeline (key stream + FlowNet/tail stream + 3 non-key lanes, hipGraph replay) on an 11-frame
    schedule with key interval 4.  For EVERY frame, as delivered by the pipeline:
      * a non-key frame's conv_feat == oracle.warp_bilinear(the aggregated feature of ITS segment's key frame,
        its own motion vectors, its own small-net feature / residual): a lane served a stale or a too-new key
        feature (hand-over ordering) fails array_equal;
      * a key frame's warp == oracle warp of the PREVIOUS key frame's feature, its choose_feat_output ==
        oracle.aggregate_softmax2 of its own taps; Proposal / PSROI head / detection NMS == oracle;
      * with conv / GEMM algorithms pinned, every tap, output and detection equals the strictly serial eager
        loop's BIT FOR BIT (so the dense stages read the right images too), and a second run of the same
        pipeline reproduces the first.
    `layout`: how the work streams were picked (FramePipeline): the timing probe's choice, a fresh stream per role, and the
    degenerate outcome of a perturbed probe — no FlowNet stream, lanes on spare streams aliased to the key queue."""
    from parity_util import assert_dets_equal, check_cur_frame, check_dets, check_key_frame, clone_dict
    from lsfa_amd.core.graphs import FrameGraphs, FramePipeline
    cfg, arg, key, cur, clip = world['cfg'], world['arg'], world['key'], world['cur'], world['clip']
    key.taps = cur.taps = None
    im_info = clip.im_info()
    # key interval 4: frames 1..11 -> key 1, cur 2-4, key 5, cur 6-8, key 9, cur 10-11
    sched = [(f, 1 + 4 * ((f - 1) // 4)) for f in range(1, 12)]
    frames = {f: clip.frame(f, DEV) for f in range(12)}
    mvs = {f: clip.motion_vector(f, kf, DEV) for f, kf in sched if f != kf}
    ress = {f: clip.res_diff(f, DEV) for f, kf in sched if f != kf}
    torch.cuda.synchronize()
    fp = FramePipeline(key, cur, cfg, H, W, DEV, lanes=3, lookahead=lookahead, taps=True, layout=layout)
    assert fp.layout_used.startswith(layout)
    if layout == "one-queue":
        assert fp.s_flow is None
    a = _run_clip_through_pipeline(fp, frames, sched, mvs, ress)
    b = _run_clip_through_pipeline(fp, frames, sched, mvs, ress)
    # the strictly serial eager loop on the same frames
    fg = FrameGraphs(key, cur, cfg, H, W, DEV, use_graphs=False, prefetch=False, taps=True)
    fg.first_frame(frames[0])
    fg.capture()
    serial = {}
    for f, kf in sched:
        if f == kf:
            d, c, _ = fg.key_frame(frames[f])
            serial[f] = dict(taps=clone_dict(fg.key_taps), out=clone_dict(fg.key_out), feat=fg.feat.clone())
        else:
            d, c, _ = fg.cur_frame(frames[f], mvs[f], ress[f])
            serial[f] = dict(taps=clone_dict(fg.cur_taps), out=clone_dict(fg.cur_out))
        serial[f].update(dets=d.clone(), counts=c.clone())
    key_feat = {0: a[0]['feat']}
    prev_key = 0
    for f, kf in sched:
        r = a[f]
        if f == kf:
            check_key_frame(cfg, r['taps'], r['out'], key_feat[prev_key], im_info)
            assert torch.equal(r['feat'], r['out']['choose_feat_output'])
            key_feat[f] = r['feat']
            prev_key = f
        else:
            check_cur_frame(cfg, arg, r['taps'], r['out'], key_feat[kf], mvs[f], ress[f], im_info)
        check_dets(cfg, r['out'], r['dets'].cpu().numpy(), r['counts'].cpu().numpy(), H, W)
    for f, _ in sched:
        for other, what in ((b, "second run of the pipeline"), (serial, "serial eager loop")):
            for name in a[f]['taps']:
                assert torch.equal(a[f]['taps'][name], other[f]['taps'][name]), "frame %d, tap %s vs %s" % (f, name, what)
            for name in a[f]['out']:
                assert torch.equal(a[f]['out'][name], other[f]['out'][name]), "frame %d, output %s vs %s" % (f, name, what)
            assert_dets_equal(a[f]['dets'], a[f]['counts'], other[f]['dets'], other[f]['counts'], "frame %d vs %s" % (f, what))
    fp.close()         # graphs and their memory pools dropped


@pytest.mark.parametrize("lookahead,group", [(False, 3), (True, 2)])
def test_frame_pipeline_batched_segments_and_key_groups(world, lookahead, group):
    """FramePipeline(segment=3, key_group=G): the three non-key frames of a segment go through the network in ONE pass (batch axis = frames,
    the reference's batch test symbol) and the fronts (backbone, FlowNet) of G consecutive key frames in one pass, 16 frames at key interval
    4: keys 1, 5, 9, 13, the last segment one frame short (per-frame lanes), the last key frame(s) without enough `upcoming` images (front
    computed alone).  For every frame as delivered: the hand-written stages equal the oracle bit for bit on that frame's own inputs - a
    non-key frame's conv_feat is the warp of ITS segment's key feature with ITS motion vectors, a key frame's warp that of the PREVIOUS key
    frame's feature; a second run reproduces the first bit for bit; a batched pass equals the same pass issued eagerly by hand (graphs,
    streams and staging add nothing); and every dense map stays within the convolution tolerance of the serial one-frame-at-a-time loop
    (a batch changes the K cut of a convolution, hence its rounding, never more)."""
    from oracle import e2e
    from parity_util import assert_dets_equal, check_cur_frame, check_dets, check_key_frame, clone_dict
    from lsfa_amd.core.graphs import FrameGraphs, FramePipeline
    from lsfa_amd.utils.synthetic import SyntheticClip
    cfg, arg, key, cur = world['cfg'], world['arg'], world['key'], world['cur']
    key.taps = cur.taps = None
    clip = SyntheticClip(3, 17, H, W, 4)
    im_info = clip.im_info()
    K, F = 4, 3
    sched = [(f, 1 + K * ((f - 1) // K)) for f in range(1, 16)]          # keys 1, 5, 9, 13; frame 16 does not exist: segment 13 has 14, 15
    keys = [f for f, kf in sched if f == kf]
    frames = {f: clip.frame(f, DEV) for f in range(16)}
    mvs = {f: clip.motion_vector(f, kf, DEV) for f, kf in sched if f != kf}
    ress = {f: clip.res_diff(f, DEV) for f, kf in sched if f != kf}
    torch.cuda.synchronize()
    predicted, sched_key = {}, dict(sched)

    def run(fp):
        outs = {}

        def keep(f, is_key):
            def deliver(bufs):
                lane = fp.delivering
                if is_key:
                    o = dict(taps=clone_dict(lane.taps), out=clone_dict(lane.out), feat=lane.feat.clone())
                else:
                    seg = getattr(bufs[0], 'lsfa_segment', None)
                    taps, out = clone_dict(lane.cur_taps), clone_dict(lane.cur_out)
                    if seg is not None:
                        taps, out = e2e.image_of_batch(taps, out, seg[1], seg[2])
                    o = dict(taps=taps, out=out, batched=seg is not None)
                    if seg is not None:      # r5: the lane a segment's pass runs on is the one stream_of_next_segment() named before its key frame was handed over
                        assert torch.cuda.current_stream(DEV) == predicted[sched_key[f]], (f, sched_key[f])
                o.update(dets=bufs[0].clone(), counts=bufs[1].clone())
                outs[f] = o
            return deliver
        first = fp.first_frame(frames[0])
        outs[0] = dict(feat=fp.feat.clone())
        predicted.clear()
        if not fp.captured:
            fp.capture()
        for f, kf in sched:
            if f == kf:
                later = [frames[k] for k in keys if k > f]
                predicted[f] = fp.stream_of_next_segment()      # where a caller would queue the upload of THIS interval's non-key inputs
                fp.key_frame(frames[f], deliver=keep(f, True), upcoming=later)
            else:
                fp.cur_frame(frames[f], mvs[f], ress[f], deliver=keep(f, False))
        fp.join()
        torch.cuda.synchronize()
        return outs

    fp = FramePipeline(key, cur, cfg, H, W, DEV, lanes=2, lookahead=lookahead, taps=True, segment=F, key_group=group, ramp=False)
    a, b = run(fp), run(fp)
    assert fp.group_sizes == ([3, 1, 3, 1] if group == 3 else [2, 2, 2, 2])
    fp.close()
    # 1. every frame against the oracle's hand-written stages, on the pipeline's own intermediate values
    key_feat, prev_key = {0: a[0]['feat']}, 0
    for f, kf in sched:
        r = a[f]
        if f == kf:
            check_key_frame(cfg, r['taps'], r['out'], key_feat[prev_key], im_info)
            assert torch.equal(r['feat'], r['out']['choose_feat_output'])
            key_feat[f], prev_key = r['feat'], f
        else:
            assert r['batched'] == (kf != 13)
            check_cur_frame(cfg, arg, r['taps'], r['out'], key_feat[kf], mvs[f], ress[f], im_info)
        check_dets(cfg, r['out'], r['dets'].cpu().numpy(), r['counts'].cpu().numpy(), H, W)
    # 2. run to run
    for f, _ in sched:
        for name in a[f]['taps']:
            assert torch.equal(a[f]['taps'][name], b[f]['taps'][name]), (f, name)
        assert_dets_equal(a[f]['dets'], a[f]['counts'], b[f]['dets'], b[f]['counts'], "frame %d, second run" % f)
    # 3. the batched passes by hand: the first full group of fronts, the first segment
    with torch.no_grad():
        grp = keys[:group]
        olds = [0] + grp[:-1]
        conv = key.key_backbone(torch.cat([frames[k] for k in grp], 0))
        flow, scale = key.key_flow(torch.cat([frames[k] for k in grp], 0), torch.cat([frames[k] for k in olds], 0))
        for i, k in enumerate(grp):
            assert torch.equal(a[k]['taps']['backbone_feat'], conv[i:i + 1]), k
            assert torch.equal(a[k]['taps']['flow'], flow[i:i + 1]) and torch.equal(a[k]['taps']['scale_map'], scale[i:i + 1]), k
        seg = [2, 3, 4]
        out = cur.forward(data=torch.cat([frames[f] for f in seg], 0), im_info=torch.from_numpy(np.repeat(im_info, F, 0)).to(DEV),
                          feat_key=key_feat[1], motion_vector=torch.cat([mvs[f] for f in seg], 0), res_diff=torch.cat([ress[f] for f in seg], 0))
        for i, f in enumerate(seg):
            assert torch.equal(a[f]['out']['conv_feat'], out['conv_feat'][i:i + 1]), f
            R = out['rois_output'].shape[0] // F
            assert torch.equal(a[f]['out']['cls_prob_reshape_output'], out['cls_prob_reshape_output'][:, i * R:(i + 1) * R]), f
    # 4. against the serial loop: same frames one at a time
    fg = FrameGraphs(key, cur, cfg, H, W, DEV, use_graphs=False, prefetch=False, taps=True)
    fg.first_frame(frames[0])
    fg.capture()
    worst = 0.0
    for f, kf in sched:
        if f == kf:
            fg.key_frame(frames[f])
            pairs = [(a[f]['feat'], fg.feat), (a[f]['taps']['backbone_feat'], fg.key_taps['backbone_feat'])]
        else:
            fg.cur_frame(frames[f], mvs[f], ress[f])
            pairs = [(a[f]['out']['conv_feat'], fg.cur_out['conv_feat']), (a[f]['taps']['small_feat'], fg.cur_taps['small_feat'])]
        for x, y in pairs:
            worst = max(worst, float((x - y).abs().max() / y.abs().max()))
    assert worst < TOL_DENSE, worst


def test_frame_pipeline_ramp_and_bank_bookkeeping(world):
    """key_group = 4 with the ramp: after first_frame (and again after flush()) the passes are 1, 2 and then 4 key fronts; a key frame
    whose front sits in the bank must hand over exactly the announced tensor (another one is an error, not a silently wrong feature);
    drop_fronts() forgets announced frames; the features do not depend on how the key frames were grouped beyond the convolutions'
    rounding (ramp on / off: within the dense tolerance)."""
    from lsfa_amd.core.graphs import FramePipeline
    from lsfa_amd.utils.synthetic import SyntheticClip
    cfg, key, cur = world['cfg'], world['key'], world['cur']
    key.taps = cur.taps = None
    clip = SyntheticClip(9, 12, H, W, 1)
    frames = [clip.frame(f, DEV) for f in range(12)]

    def run(ramp, upto=9):
        fp = FramePipeline(key, cur, cfg, H, W, DEV, lanes=2, use_graphs=False, key_group=4, ramp=ramp)
        fp.first_frame(frames[0])
        fp.capture()
        feats = []
        for k in range(1, upto):
            fp.key_frame(frames[k], upcoming=frames[k + 1:upto])
            feats.append(fp.feat.clone())
        fp.join()
        torch.cuda.synchronize()
        return fp, feats
    fp, with_ramp = run(True)
    assert fp.group_sizes == [1, 2, 4, 1]                      # 8 key frames: 1 + 2 + 4, then one left with nothing ahead
    fp.flush()
    fp.key_frame(frames[9], upcoming=frames[10:12])
    assert fp.group_sizes[-1] == 1                             # the pipeline ran empty: the ramp starts over
    fp.key_frame(frames[10], upcoming=frames[11:12])
    assert fp.group_sizes[-1] == 2
    with pytest.raises(ValueError, match="another image"):
        fp.key_frame(frames[3])                                # frames[11]'s front is what the bank holds
    assert len(fp._bank_ready) == 1                            # the refused call left the bank as it was
    fp.drop_fronts()
    fp.key_frame(frames[3])                                    # fine now: computed alone
    fp.join()
    torch.cuda.synchronize()
    fp.close()
    fp2, without = run(False)
    assert fp2.group_sizes == [4, 4]
    fp2.close()
    worst = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(with_ramp, without))
    assert worst < TOL_DENSE, worst


def test_frame_pipeline_key_groups_above_six_keep_few_banks(world):
    """key_group = 8 (r6: bench.py runs 12): banks exist for group sizes 2..6 and the full size only, so a run of 15 key frames is computed
    as passes of 8, 6 and 1 fronts - the tail takes the largest size there is a bank for - and the aggregated features are those of
    key_group = 4 up to the convolutions' rounding (a different batch changes a launch's K cut and shared scale, never more)."""
    from lsfa_amd.core.graphs import FramePipeline
    from lsfa_amd.utils.synthetic import SyntheticClip
    cfg, key, cur = world['cfg'], world['key'], world['cur']
    key.taps = cur.taps = None
    clip = SyntheticClip(21, 16, H, W, 1)
    frames = [clip.frame(f, DEV) for f in range(16)]

    def run(group):
        fp = FramePipeline(key, cur, cfg, H, W, DEV, lanes=2, use_graphs=False, key_group=group)
        assert sorted(fp.banks) == ([2, 3, 4, 5, 6, 8] if group == 8 else [2, 3, 4])
        fp.first_frame(frames[0])
        fp.capture()
        feats = []
        for k in range(1, 16):
            fp.key_frame(frames[k], upcoming=frames[k + 1:16])
            fp.join()
            torch.cuda.synchronize()
            feats.append(fp.feat.clone())
        sizes = list(fp.group_sizes)
        fp.close()
        return sizes, feats
    sizes8, f8 = run(8)
    assert sizes8 == [8, 6, 1], sizes8
    sizes4, f4 = run(4)
    assert sizes4 == [4, 4, 4, 3], sizes4
    worst = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(f8, f4))
    assert worst < TOL_DENSE, worst


def test_frame_pipeline_batched_with_two_clips_in_lockstep(world):
    """FramePipeline(batch=2, segment=3, key_group=2): two clips advance together, so a segment pass carries 3 frames x 2 clips (frame-major:
    image f * 2 + b samples clip b's key feature - lsfa_warp_bilinear's feat_n = 2 < N = 6) and a key group 2 key frames x 2 clips.  Every
    delivered frame equals the same passes issued by hand (bit for bit), and for one key frame and one non-key frame each clip's image, cut
    out of its batch, passes the oracle's hand-written-stage checks against ITS OWN clip's key feature."""
    from oracle import e2e
    from parity_util import check_cur_frame, check_key_frame, clone_dict
    from lsfa_amd.core.graphs import FramePipeline
    from lsfa_amd.utils.synthetic import SyntheticClip
    cfg, arg, key, cur = world['cfg'], world['arg'], world['key'], world['cur']
    key.taps = cur.taps = None
    B, K, F = 2, 4, 3
    clips = [SyntheticClip(5 + b, 10, H, W, K) for b in range(B)]
    cat = lambda fn: torch.cat([fn(c) for c in clips], 0)
    im_info = clips[0].im_info()
    sched = [(f, 1 + K * ((f - 1) // K)) for f in range(1, 9)]           # keys 1, 5; segments 2-4, 6-8
    keys = [1, 5]
    frames = {f: cat(lambda c: c.frame(f, DEV)) for f in range(9)}
    mvs = {f: cat(lambda c: c.motion_vector(f, kf, DEV)) for f, kf in sched if f != kf}
    ress = {f: cat(lambda c: c.res_diff(f, DEV)) for f, kf in sched if f != kf}
    fp = FramePipeline(key, cur, cfg, H, W, DEV, lanes=2, taps=True, batch=B, segment=F, key_group=2)       # (no ramp below key_group 3)
    outs = {}

    def keep(f, is_key):
        def deliver(bufs):
            lane = fp.delivering
            if is_key:
                outs[f] = dict(taps=clone_dict(lane.taps), out=clone_dict(lane.out), feat=lane.feat.clone(), dets=bufs[0].clone(), counts=bufs[1].clone())
            else:
                _, i, n = bufs[0].lsfa_segment
                outs[f] = dict(taps=clone_dict(lane.cur_taps), out=clone_dict(lane.cur_out), index=i, dets=bufs[0].clone(), counts=bufs[1].clone())
        return deliver
    fp.first_frame(frames[0])
    feat0 = fp.feat.clone()
    fp.capture()
    for f, kf in sched:
        if f == kf:
            fp.key_frame(frames[f], deliver=keep(f, True), upcoming=[frames[k] for k in keys if k > f])
        else:
            fp.cur_frame(frames[f], mvs[f], ress[f], deliver=keep(f, False))
    fp.join()
    torch.cuda.synchronize()
    fp.close()
    assert outs[1]['dets'].shape[0] == B and outs[2]['dets'].shape[0] == B
    im_t = torch.from_numpy(np.repeat(im_info, B, 0)).to(DEV)
    with torch.no_grad():
        conv = key.key_backbone(torch.cat([frames[1], frames[5]], 0))
        flow, scale = key.key_flow(torch.cat([frames[1], frames[5]], 0), torch.cat([frames[0], frames[1]], 0))
        feat, prev = {}, feat0
        for i, k in enumerate(keys):
            sl = slice(i * B, (i + 1) * B)
            assert torch.equal(outs[k]['taps']['backbone_feat'], conv[sl]) and torch.equal(outs[k]['taps']['flow'], flow[sl])
            feat[k] = key.key_aggregate(conv[sl], flow[sl], scale[sl], prev)
            assert torch.equal(outs[k]['feat'], feat[k]), k
            prev = feat[k]
        for k in keys:
            seg = [k + 1, k + 2, k + 3]
            out = cur.forward(data=torch.cat([frames[f] for f in seg], 0), im_info=im_t.repeat(F, 1), feat_key=feat[k],
                              motion_vector=torch.cat([mvs[f] for f in seg], 0), res_diff=torch.cat([ress[f] for f in seg], 0))
            for i, f in enumerate(seg):
                assert outs[f]['index'] == i
                assert torch.equal(outs[f]['out']['conv_feat'], out['conv_feat']), f          # the lane's whole batch, cloned at every delivery
    # each clip's image against the oracle's stages, with its own clip's key feature
    for b in range(B):
        t, o = e2e.image_of_batch(outs[5]['taps'], outs[5]['out'], b, B)
        check_key_frame(cfg, t, o, outs[1]['feat'][b:b + 1], im_info)
        i = 1 * B + b                                                                     # frame 7 = index 1 of segment 6-8
        t, o = e2e.image_of_batch(outs[7]['taps'], outs[7]['out'], i, F * B)
        check_cur_frame(cfg, arg, t, o, outs[5]['feat'][b:b + 1], mvs[7][b:b + 1], ress[7][b:b + 1], im_info)


def test_pred_eval_pipelined_two_videos(world):
    """Two videos of the same shape through pred_eval_pipelined: the second video reuses the first
    one's captured pipeline after a drain; frame ids and every detection row equal the serial
    pred_eval's exactly (conv / GEMM algorithms pinned)."""
    from lsfa_amd.config.config import lsfa_test_config
    from lsfa_amd.function.test_rcnn import test_rcnn
    from lsfa_amd.utils.synthetic import synthetic_roidb
    cfg = lsfa_test_config(key_frame_interval=3)
    arg, aux = world['arg'], world['aux']
    roidb = synthetic_roidb(2, 8, H, W, 3)
    rows_s, ids_s = test_rcnn(cfg, roidb, arg, aux, device=DEV, pipeline=False)
    rows_p, ids_p = test_rcnn(cfg, roidb, arg, aux, device=DEV, pipeline=True)
    np.testing.assert_array_equal(ids_s, ids_p)
    assert len(rows_s) > 0
    np.testing.assert_array_equal(rows_s, rows_p)


def test_pred_eval_pipelined_with_batched_passes(world):
    """pred_eval_pipelined(segment = interval - 1, key_group = 3) - the reference-shaped frame loop on the batched pipeline, the coming key
    frames' images obtained from the loader (TestLoader.upcoming_key_frames: the SAME tensors come back when the iteration reaches them) -
    over two videos of 14 frames at key interval 4 (three full segments, a short one before the video's last frame, which is a key frame
    by the loader's rule; passes of up to 3 key fronts): the same frame ids as the serial pred_eval, and its detections up
    to the rounding of the convolutions' different K cuts - per (frame, class) the same number of rows wherever no score sits on the
    threshold, scores within 1e-4 (north_star's tolerance), boxes within 0.01 px."""
    from lsfa_amd.config.config import lsfa_test_config
    from lsfa_amd.function.test_rcnn import test_rcnn
    from lsfa_amd.utils.synthetic import synthetic_roidb
    cfg = lsfa_test_config(key_frame_interval=4)
    arg, aux = world['arg'], world['aux']
    roidb = synthetic_roidb(2, 14, H, W, 4)
    rows_s, ids_s = test_rcnn(cfg, roidb, arg, aux, device=DEV, pipeline=False)
    rows_b, ids_b = test_rcnn(cfg, roidb, arg, aux, device=DEV, pipeline=True, segment=3, key_group=3)
    np.testing.assert_array_equal(ids_s, ids_b)
    assert len(rows_s) > 0 and abs(len(rows_b) - len(rows_s)) <= 0.01 * len(rows_s)

    def cells(rows):
        out = {}
        for r in rows:
            out.setdefault((int(r[0]), int(r[1])), []).append(r[2:])
        return {k: np.asarray(sorted(v, key=lambda x: -x[0])) for k, v in out.items()}
    a, b = cells(rows_s), cells(rows_b)
    same = [k for k in a if k in b and len(a[k]) == len(b[k])]
    assert len(same) >= 0.98 * len(a), (len(same), len(a), len(b))
    dscore = max(float(np.abs(a[k][:, 0] - b[k][:, 0]).max()) for k in same)
    assert dscore <= 1e-4, dscore
    # boxes of the rows whose scores pair up unambiguously (scores of one cell further apart than the tolerance)
    dbox = 0.0
    for k in same:
        s = a[k][:, 0]
        if len(s) < 2 or float(np.min(-np.diff(s))) > 2e-4:
            dbox = max(dbox, float(np.abs(a[k][:, 1:] - b[k][:, 1:]).max()))
    assert dbox <= 1e-2, dbox


def test_pred_eval_pipelined_evicts_pipelines_of_old_shapes(world):
    """Five videos of three frame shapes (A, B, C, A, B) with room for TWO captured pipelines: the oldest shape's pipeline is closed (graphs
    and their memory pools dropped, streams parked) when a third shape arrives and rebuilt when its shape returns; every detection row
    equals the serial pred_eval's, and the device memory in use after the run is what it was before (no pool of an evicted pipeline
    survives) - ADVICE r3."""
    from lsfa_amd.config.config import lsfa_test_config
    from lsfa_amd.core import streams
    from lsfa_amd.function.test_rcnn import test_rcnn
    from lsfa_amd.utils.synthetic import synthetic_roidb
    cfg = lsfa_test_config(key_frame_interval=3)
    arg, aux = world['arg'], world['aux']
    shapes = [(128, 192), (160, 256), (96, 160), (128, 192), (160, 256)]
    roidb, fid = [], 0
    for v, (h, w) in enumerate(shapes):
        e = synthetic_roidb(1, 5, h, w, 3, seed=v)[0]
        e['frame_id'] = fid
        fid += 5
        roidb.append(e)
    rows_s, ids_s = test_rcnn(cfg, roidb, arg, aux, device=DEV, pipeline=False)
    torch.cuda.synchronize()
    gc.collect()
    torch.cuda.empty_cache()
    before, parked = torch.cuda.memory_allocated(), sum(len(v) for v in streams._FREE.values())
    rows_p, ids_p = test_rcnn(cfg, roidb, arg, aux, device=DEV, pipeline=True, max_pipelines=2)
    torch.cuda.synchronize()
    gc.collect()                     # lanes, deliver closures and graphs reference each other: the pools go when the cycle is collected
    torch.cuda.empty_cache()
    after = torch.cuda.memory_allocated()
    np.testing.assert_array_equal(ids_s, ids_p)
    assert len(rows_s) > 0
    np.testing.assert_array_equal(rows_s, rows_p)
    assert after <= before + (64 << 20), (before, after)
    # five pipelines were built (A, B, C, A again, B again), every one closed: their streams are parked, a bounded number each (22 today)
    grown = sum(len(v) for v in streams._FREE.values()) - parked
    assert 5 <= grown <= 5 * 32, grown


def test_two_clips_interleaved_on_one_gpu_are_isolated(world):
    """BASELINE configs[2]/[3] run several independent clips per process / node.  Two clips pushed
    alternately through two FramePipelines on the same GPU (bf16 contractions, config 3's arithmetic)
    give bit-for-bit what each gives alone: pipelines share no buffers, workspaces or events."""
    from lsfa_amd.core.graphs import FramePipeline
    from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
    from lsfa_amd.utils.synthetic import SyntheticClip
    cfg, arg, aux = world['cfg'], world['arg'], world['aux']
    net = resnet_v1_101_flownet_rfcn(cfg)
    key = net.get_key_test_symbol(cfg).bind(arg, aux, DEV, torch.bfloat16)
    cur = net.get_cur_test_symbol(cfg).bind(arg, aux, DEV, torch.bfloat16)
    clips = [SyntheticClip(c, 8, H, W) for c in (3, 4)]
    sched = [(f, 1 + 3 * ((f - 1) // 3)) for f in range(1, 8)]        # key 1, cur 2-3, key 4, cur 5-6, key 7
    inputs = [{f: (c.frame(f, DEV), None if f == kf else c.motion_vector(f, kf, DEV), None if f == kf else c.res_diff(f, DEV))
               for f, kf in sched} for c in clips]
    first = [c.frame(0, DEV) for c in clips]
    torch.cuda.synchronize()
    fps = [FramePipeline(key, cur, cfg, H, W, DEV) for _ in clips]

    def run(order):
        outs = [{}, {}]
        for ci in set(order):
            fps[ci].first_frame(first[ci])
            if not fps[ci].captured:
                fps[ci].capture()
        for f, kf in sched:
            for ci in order:
                data, mv, res = inputs[ci][f]
                keep = (lambda ci, f: (lambda b: outs[ci].__setitem__(f, (b[0].clone(), b[1].clone()))))(ci, f)
                if f == kf:
                    fps[ci].key_frame(data, deliver=keep)
                else:
                    fps[ci].cur_frame(data, mv, res, deliver=keep)
        for ci in set(order):
            fps[ci].join()
        torch.cuda.synchronize()
        return [{f: (d.cpu().numpy(), c.cpu().numpy()) for f, (d, c) in o.items()} for o in outs]

    run([0, 1])                          # throw-away pass
    alone = [run([0])[0], run([1])[1]]
    both = run([0, 1])
    for ci in (0, 1):
        for f, _ in sched:
            np.testing.assert_array_equal(both[ci][f][1], alone[ci][f][1])
            np.testing.assert_array_equal(both[ci][f][0], alone[ci][f][0])
    assert any((both[0][f][0] != both[1][f][0]).any() for f, _ in sched)       # the clips do differ


def test_bench_line_contract():
    """bench.py prints ONE JSON line with the contract's keys (small frame size so it is quick)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--height", "192",
                          "--width", "320", "--cpu-budget-s", "5", "--settle-s", "0.3"], capture_output=True, text=True, timeout=600,
                         cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "frames/s" and d["value"] > 0
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]           # the dominant hand-written kernel: the split-bf16 convolution, priced against the matrix pipe
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launches"] > 0 and r["traffic"] is None and 2500.0 / 6 - 0.1 <= r["peak"] <= 2500.0 / 3 + 0.1
    h = d["roofline_hbm_kernel"]  # the HBM-bound one (warp)
    assert h["bound"] == "hbm" and h["unit"] == "GB/s" and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "frames/s" and c["sample"]
    # r3: three back-to-back timed regions (`value` is the first), what ran before the warm-up steps, exit status 0 = parity held
    v = d["value_spread"]
    assert v["repeats"] == 3 and len(v["values"]) == 3 and v["values"][0] == d["value"] and v["min"] <= d["value"] <= v["max"]
    assert "settle" in d["config"]["setup_before_warmup"] and d["warmup"] == 1
    assert d["parity"]["handwritten_stage_mismatches_on_gpu_inputs"] == 0


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_frames_run_without_a_library_convolution(world, monkeypatch, dtype):
    """Neither the fp32 nor the bf16 key / non-key graphs reach a library convolution OR a library GEMM: every convolution runs on
    lsfa_conv_fwd (fp32: two fp16 pieces per operand; bf16: one bf16 piece - r3's bf16 mode fell back to F.conv2d / MIOpen), the R-FCN
    score maps and the RPN head (straight on the NCHW map) are among them (r4: torch.mm / addmm / matmul / softmax are forbidden here too).
    MIOpen picks its solver from per-user state that concurrently starting processes race for, which made detections depend on the rank
    layout; hipBLASLt keeps a workspace per stream that a pipeline's captured graphs pin for the life of the
    process.  The executors' status words stay clear (no fp16 scale overflowed)."""
    import torch.nn.functional as F
    cfg, clip = world['cfg'], world['clip']
    if dtype == "f32":
        key, cur = world['key'], world['cur']
    else:
        from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
        net = resnet_v1_101_flownet_rfcn(cfg)
        key = net.get_key_test_symbol(cfg).bind(world['arg'], world['aux'], DEV, torch.bfloat16)
        cur = net.get_cur_test_symbol(cfg).bind(world['arg'], world['aux'], DEV, torch.bfloat16)
    im_info_t = torch.from_numpy(clip.im_info()).to(DEV)
    f0, f3, f10 = clip.frame(0).to(DEV), clip.frame(3).to(DEV), clip.frame(10).to(DEV)

    def forbidden(*a, **k):
        raise AssertionError("a library convolution was called")
    for name in ("conv2d", "conv_transpose2d", "conv1d", "conv3d", "max_pool2d", "avg_pool2d", "unfold", "linear", "softmax"):
        monkeypatch.setattr(F, name, forbidden)
    for name in ("mm", "addmm", "matmul", "bmm", "baddbmm", "einsum", "softmax"):
        monkeypatch.setattr(torch, name, forbidden)
    out0 = key.forward(data=f0, im_info=im_info_t, data_key_old=f0, feat_key_old=torch.zeros(1, 1024, 1, 1, device=DEV))
    feat0 = out0['choose_feat_output']
    cur.forward(data=f3, im_info=im_info_t, feat_key=feat0, motion_vector=clip.motion_vector(3, 0).to(DEV), res_diff=clip.res_diff(3).to(DEV))
    out10 = key.forward(data=f10, im_info=im_info_t, data_key_old=f0, feat_key_old=feat0)
    torch.cuda.synchronize()
    assert out10['rois_output'].shape == (300, 5)
    assert bool(torch.isfinite(out10['choose_feat_output']).all())
    key.check_status()
    cur.check_status()
