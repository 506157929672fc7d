"""BASELINE configs[1] at its real size: 1000x600 frames (38x63 feature map, 21,546 anchors, 300 ROIs)
through the key / cur graphs vs oracle/graph_ref.py.

Dense stages (library MFMA convolutions / GEMMs, BN folded, channels-last) are compared with the
unfused torch-CPU statement and the measured error is written to gpurun_out/parity_fullres.json;
every hand-written stage (flow/MV warp with its fused epilogues, Nq softmax combine, Proposal,
PSROI + average + softmax, detection post-processing) is pinned to the oracle BIT FOR BIT on
the GPU's own inputs to that stage.  Finally the un-forced end-to-end outputs are judged against
a FLOAT64 statement of the same graph (oracle/e2e.py): the GPU path may be at most 1.5x as far from
float64 as the fp32 oracle is (RPN scores / deltas, ROI coordinates, decoded boxes, class
probabilities), ROIs are identified by ANCHOR INDEX, and any pair of proposals the two fp32 sides
order differently must be a tie in float64 (scores closer than the fp32 error bar) - otherwise the
test fails.  No constant bounds the GPU-vs-oracle distance itself (r3 had 1e-3 px and "up to 4
displaced rows" here).
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import e2e, graph_ref
from parity_util import check_cur_frame, check_dets, check_heads, check_key_frame, np_, record, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H, W = 600, 1000
# dense contractions, fp32 MFMA vs torch-CPU fp32 of the unfused graph (different summation orders,
# BN folded in fp64 vs applied in fp32): relative to the map's max.  Measured r2: see DESIGN.md §5.
TOL_DENSE = 2e-5
# end to end, each side on its own values (no teacher forcing): class probabilities within north_star's 1e-4; everything else
# is judged relative to the float64 graph by oracle/e2e.py
TOL_SCORE = 1e-4
F64 = torch.float64
TOL_BOX_PX_BACKSTOP = 1e-3          # px, GPU vs fp32 oracle (reported 2-4e-4)
ORACLE_BOX_ERR_CEILING = 1e-3       # px, fp32 oracle vs float64 (measured 2.4e-4): the yardstick of the relative criterion may not drift


@pytest.fixture(scope="module")
def world():
    from lsfa_amd.config.config import lsfa_test_config
    from lsfa_amd.symbols import params as P
    from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
    from lsfa_amd.utils.synthetic import SyntheticClip
    cfg = lsfa_test_config(key_frame_interval=10)
    arg, aux = P.init_params(cfg, seed=0)
    net = resnet_v1_101_flownet_rfcn(cfg)
    key = net.get_key_test_symbol(cfg).bind(arg, aux, DEV)
    cur = net.get_cur_test_symbol(cfg).bind(arg, aux, DEV)
    clip = SyntheticClip(0, 12, H, W)
    return dict(cfg=cfg, arg=arg, aux=aux, key=key, cur=cur, clip=clip)


def test_first_key_cur_second_key_at_1000x600(world):
    cfg, arg, aux, key, cur, clip = (world[k] for k in ('cfg', 'arg', 'aux', 'key', 'cur', 'clip'))
    im_info = clip.im_info()
    im_t = torch.from_numpy(im_info).to(DEV)
    f0, f3, f10 = clip.frame(0), clip.frame(3), clip.frame(10)
    p = graph_ref.Params(arg, aux)
    rec = {}

    # ---- frame 0 (flag 0): backbone + heads -----------------------------------------------
    key.taps = {}
    out0 = key.forward(data=f0.to(DEV), im_info=im_t, data_key_old=f0.to(DEV), feat_key_old=torch.zeros(1, 1024, 1, 1, device=DEV))
    taps0 = dict(key.taps)
    ref0 = graph_ref.key_forward(cfg, arg, aux, f0.numpy(), f0.numpy(), np.zeros((1, 1024, 1, 1), np.float32), im_info)
    assert out0['choose_feat_output'].shape == (1, 1024, 38, 63)
    rec['backbone_feat'] = rel_err(np_(taps0['backbone_feat']), ref0['backbone_feat'])
    rec['cls_map'] = rel_err(np_(taps0['cls_map']), ref0['cls_map'])
    rec['box_map'] = rel_err(np_(taps0['box_map']), ref0['box_map'])
    rec['rpn_bbox_pred'] = rel_err(np_(taps0['rpn_bbox_pred']), ref0['rpn_bbox_pred'])
    rec['rpn_cls_prob_abs'] = float(np.abs(np_(taps0['rpn_cls_prob']) - ref0['rpn_cls_prob']).max())
    check_heads(cfg, taps0, out0, im_info)
    from lsfa_amd import hip
    d, c, _ = hip.det_postprocess(out0['rois_output'], out0['bbox_pred_reshape_output'][0], out0['cls_prob_reshape_output'][0],
                                  H, W, 1.0, nms_thresh=cfg.TEST.NMS, max_per_image=cfg.TEST.max_per_image)
    check_dets(cfg, out0, d.cpu().numpy(), c.cpu().numpy(), H, W)
    d0 = graph_ref.key_forward(cfg, arg, aux, f0.numpy(), f0.numpy(), np.zeros((1, 1024, 1, 1), np.float32), im_info, dtype=F64)
    rec['frame0_end_to_end'] = e2e.frame_gap(cfg, e2e.gpu_side(cfg, taps0, out0, im_info), ref0, d0, im_info, H, W)
    rec['backbone_feat_vs_f64'] = dict(gpu=rel_err(np_(taps0['backbone_feat']), d0['backbone_feat']), oracle_fp32=rel_err(ref0['backbone_feat'], d0['backbone_feat']))
    feat0 = out0['choose_feat_output']

    # ---- frame 3 (flag 2): small net + MV warp + residual + heads ---------------------------
    mv, res = clip.motion_vector(3, 0), clip.res_diff(3)
    cur.taps = {}
    out3 = cur.forward(data=f3.to(DEV), im_info=im_t, feat_key=feat0, motion_vector=mv.to(DEV), res_diff=res.to(DEV))
    taps3 = dict(cur.taps)
    rec['small_feat'] = rel_err(np_(taps3['small_feat']), graph_ref.small_net_feature(p, f3).numpy())
    check_cur_frame(cfg, arg, taps3, out3, feat0, mv, res, im_info)
    ref3 = graph_ref.cur_forward(cfg, arg, aux, f3.numpy(), ref0['choose_feat_output'], mv.numpy(), res.numpy(), im_info)
    d3 = graph_ref.cur_forward(cfg, arg, aux, f3.numpy(), d0['choose_feat_output'], mv.numpy(), res.numpy(), im_info, dtype=F64)
    rec['frame3_end_to_end'] = e2e.frame_gap(cfg, e2e.gpu_side(cfg, taps3, out3, im_info), ref3, d3, im_info, H, W)

    # ---- frame 10 (flag 1): FlowNet + flow warp x scale map + Nq aggregation + heads ---------
    key.taps = {}
    out10 = key.forward(data=f10.to(DEV), im_info=im_t, data_key_old=f0.to(DEV), feat_key_old=feat0)
    taps10 = dict(key.taps)
    flow_ref, scale_ref = graph_ref.get_flownet(p, f10, f0)
    rec['flow_abs'] = float(np.abs(np_(taps10['flow']) - flow_ref.numpy()).max())
    rec['flow_max'] = float(flow_ref.abs().max())
    rec['scale_map'] = rel_err(np_(taps10['scale_map']), scale_ref.numpy())
    check_key_frame(cfg, taps10, out10, feat0, im_info)
    logits_ref = graph_ref.nq_logits(p, torch.from_numpy(np_(taps10['warp'])), taps10['backbone_feat'].cpu()).numpy()
    rec['nq_logits_abs'] = float(np.abs(np_(taps10['nq_logits']) - logits_ref).max())
    rec['nq_logits_max'] = float(np.abs(logits_ref).max())
    ref10 = graph_ref.key_forward(cfg, arg, aux, f10.numpy(), f0.numpy(), ref0['choose_feat_output'], im_info)
    rec['choose_feat_second_key'] = rel_err(np_(out10['choose_feat_output']), ref10['choose_feat_output'])
    d10 = graph_ref.key_forward(cfg, arg, aux, f10.numpy(), f0.numpy(), d0['choose_feat_output'], im_info, dtype=F64)
    rec['frame10_end_to_end'] = e2e.frame_gap(cfg, e2e.gpu_side(cfg, taps10, out10, im_info), ref10, d10, im_info, H, W)
    rec['choose_feat_second_key_vs_f64'] = dict(gpu=rel_err(np_(out10['choose_feat_output']), d10['choose_feat_output']),
                                                oracle_fp32=rel_err(ref10['choose_feat_output'], d10['choose_feat_output']))
    # how deep the Proposal's sweep goes on this data (what a mask limited to the first rows could save): the position
    # of the 300th survivor in the score-sorted list of 6000, per frame, from the oracle's own sweep on the GPU's RPN outputs
    depth = {}
    for name, taps in (('frame0', taps0), ('frame3', taps3), ('frame10', taps10)):
        _, _, _, keep, nkeep = oracle.proposal(np_(taps['rpn_cls_prob']), np_(taps['rpn_bbox_pred']), im_info, rpn_min_size=cfg.TEST.RPN_MIN_SIZE, return_debug=True)
        n = int(nkeep[0])                                          # the oracle sweeps the whole list; the operator stops at post_nms_top_n = 300
        last = int(keep[0][min(n, cfg.TEST.RPN_POST_NMS_TOP_N) - 1])
        depth[name] = dict(survivors_of_the_full_sweep=n, position_of_survivor_300=last, blocks_of_64_visited=last // 64 + 1, blocks_total=(6000 + 63) // 64)
    rec['proposal_sweep_depth'] = depth
    key.taps = cur.taps = None
    record('fullres', rec)
    world['oracles'] = dict(feat0=feat0, ref3=ref3, d3=d3, ref10=ref10, d10=d10)       # for the batched-pass test below
    world['frame0'] = dict(taps=taps0, out=out0, ref=ref0, d64=d0)                      # for the exact-fp32 A/B below

    for k in ('backbone_feat', 'cls_map', 'box_map', 'rpn_bbox_pred', 'small_feat', 'scale_map', 'choose_feat_second_key'):
        assert rec[k] < TOL_DENSE, (k, rec[k])
    assert rec['rpn_cls_prob_abs'] < 1e-4
    assert rec['flow_abs'] < TOL_DENSE * max(1.0, rec['flow_max'])
    assert rec['nq_logits_abs'] < TOL_DENSE * max(1.0, rec['nq_logits_max'])
    for k in ('frame0_end_to_end', 'frame3_end_to_end', 'frame10_end_to_end'):
        e = rec[k]
        assert not e['failures'], (k, e['failures'], e)       # the float64-anchored criterion: oracle/e2e.py (a)-(d)
        assert e['max_abs_dscore'] <= TOL_SCORE, (k, e)         # north_star's tolerance on the class probabilities
        assert e['rois_compared'] >= 250, (k, e)                # the three graphs kept (nearly) the same proposals: the sample is the frame
        # absolute backstops next to the relative criterion: a degraded ORACLE must not widen what the GPU may do.  The fp32
        # oracle's own distance to float64 is bounded, and the GPU-vs-oracle box distance stays under a loose constant (10 x north_star's
        # 1e-4 px: a decoded corner of a 1000-px box carries a few fp32 ulps, 6.1e-5 px each, on either side)
        assert e['max_abs_dbox'] <= TOL_BOX_PX_BACKSTOP, (k, e['max_abs_dbox'])
        assert e['err_vs_f64_box_px']['oracle_fp32'] <= ORACLE_BOX_ERR_CEILING, (k, e['err_vs_f64_box_px'])
    # the dense features against float64: the GPU (split operands on the fp16 / bf16 matrix pipe) no further than 1.5x the fp32 oracle
    for k in ('backbone_feat_vs_f64', 'choose_feat_second_key_vs_f64'):
        assert rec[k]['gpu'] <= e2e.RATIO * rec[k]['oracle_fp32'] + 2.0 ** -23, (k, rec[k])


def test_batched_passes_meet_the_same_criterion_at_1000x600(world):
    """What FramePipeline(segment=9, key_group=3) computes - the fronts of three key frames in one pass, the nine non-key frames of a
    segment in one pass - judged like the frame-by-frame graphs above: key frame 10 (slot 0 of a group with two more images) and non-key
    frame 3 (image 2 of the segment 1..9), each cut out of its batch, against the SAME fp32 and float64 oracle frames.  A batch changes a
    convolution's launch plan and with it the summation order; the float64-anchored criterion is what says that this is as good an fp32
    evaluation as the oracle's."""
    if 'oracles' not in world:
        pytest.skip("runs after test_first_key_cur_second_key_at_1000x600 (its oracle frames are reused)")
    cfg, arg, key, cur, clip = (world[k] for k in ('cfg', 'arg', 'key', 'cur', 'clip'))
    o = world['oracles']
    im_info = clip.im_info()
    im_t = torch.from_numpy(im_info).to(DEV)
    feat0 = o['feat0']
    rec = {}
    with torch.no_grad():
        group, olds = [clip.frame(f).to(DEV) for f in (10, 11, 9)], [clip.frame(f).to(DEV) for f in (0, 10, 11)]
        conv = key.key_backbone(torch.cat(group, 0))
        flow, scale = key.key_flow(torch.cat(group, 0), torch.cat(olds, 0))
        key.taps = {}
        feat10 = key.key_aggregate(conv[0:1], flow[0:1], scale[0:1], feat0)
        out10 = key.key_heads(feat10, im_t)
        taps10 = dict(key.taps, backbone_feat=conv[0:1])
        key.taps = None
        check_key_frame(cfg, taps10, out10, feat0, im_info)
        rec['frame10_in_a_group_of_three'] = e2e.frame_gap(cfg, e2e.gpu_side(cfg, taps10, out10, im_info), o['ref10'], o['d10'], im_info, H, W)
        rec['choose_feat_vs_f64'] = dict(gpu=rel_err(np_(feat10), o['d10']['choose_feat_output']),
                                         oracle_fp32=rel_err(o['ref10']['choose_feat_output'], o['d10']['choose_feat_output']))
        seg = list(range(1, 10))
        mvs, ress = [clip.motion_vector(f, 0) for f in seg], [clip.res_diff(f) for f in seg]
        cur.taps = {}
        outF = cur.forward(data=torch.cat([clip.frame(f).to(DEV) for f in seg], 0), im_info=im_t.repeat(len(seg), 1), feat_key=feat0,
                           motion_vector=torch.cat(mvs, 0).to(DEV), res_diff=torch.cat(ress, 0).to(DEV))
        tapsF = dict(cur.taps)
        cur.taps = None
        t3, o3 = e2e.image_of_batch(tapsF, outF, 2, len(seg))
        check_cur_frame(cfg, arg, t3, o3, feat0, mvs[2], ress[2], im_info)
        rec['frame3_in_a_segment_of_nine'] = e2e.frame_gap(cfg, e2e.gpu_side(cfg, t3, o3, im_info), o['ref3'], o['d3'], im_info, H, W)
    key.check_status()
    cur.check_status()
    record('fullres_batched', rec)
    for k in ('frame10_in_a_group_of_three', 'frame3_in_a_segment_of_nine'):
        e = rec[k]
        assert not e['failures'], (k, e['failures'], e)
        assert e['max_abs_dscore'] <= TOL_SCORE, (k, e)
        assert e['rois_compared'] >= 250, (k, e)
    assert rec['choose_feat_vs_f64']['gpu'] <= e2e.RATIO * rec['choose_feat_vs_f64']['oracle_fp32'] + 2.0 ** -23, rec['choose_feat_vs_f64']


def test_map_agreement_at_1000x600_in_both_dtypes(world):
    """north_star: "mAP ... within 0.1 of the reference".  Frame 0 + one ten-frame interval at 1000x600 through the eager frame loop in fp32
    (two fp16 pieces per product) AND in the bf16 mode (BASELINE configs[2]: one bf16 product), against the fp32 oracle's rows of the same
    frames: VID mAP@0.5 with the oracle's five most confident detections per frame as ground truth (oracle/map_check.py).  r4 asserted this
    at 192x320 in fp32 only; the bf16 mode's criterion was "feature within 3 %"."""
    from oracle import map_check
    from lsfa_amd.core.graphs import FrameGraphs
    from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
    cfg, arg, aux, clip = (world[k] for k in ('cfg', 'arg', 'aux', 'clip'))
    K = 10
    im_info = clip.im_info()
    fr = lambda f: clip.frame(f).numpy()
    rows_ref = map_check.oracle_rows(cfg, arg, aux, fr, lambda f: clip.motion_vector(f, 1).numpy(), lambda f: clip.res_diff(f).numpy(), im_info, K, 1)
    net = resnet_v1_101_flownet_rfcn(cfg)
    rec = {}
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        key = world['key'] if dt == torch.float32 else net.get_key_test_symbol(cfg).bind(arg, aux, DEV, dtype=dt)
        cur = world['cur'] if dt == torch.float32 else net.get_cur_test_symbol(cfg).bind(arg, aux, DEV, dtype=dt)
        eg = FrameGraphs(key, cur, cfg, H, W, DEV, use_graphs=False, prefetch=False, batch=1)
        eg.first_frame(clip.frame(0, DEV))
        eg.capture()
        rows = []
        for f in range(1, K + 1):
            if f == 1:
                d, c, _ = eg.key_frame(clip.frame(f, DEV))
            else:
                d, c, _ = eg.cur_frame(clip.frame(f, DEV), clip.motion_vector(f, 1, DEV), clip.res_diff(f, DEV))
            d, c = d.cpu().numpy(), c.cpu().numpy()
            for j in range(1, d.shape[0]):
                for k in range(int(c[j])):
                    rows.append([f, j, d[j, k, 4]] + list(d[j, k, :4]))
        key.check_status()
        cur.check_status()
        m = map_check.map_vs_oracle(np.asarray(rows, np.float64).reshape(-1, 7), rows_ref, range(1, K + 1), cfg.dataset.NUM_CLASSES)
        rec[name] = m
        assert m["map_oracle"] > 0.5, m
        assert abs(m["delta"]) < 0.1, (name, m)
    assert abs(rec["f32"]["delta"]) < 0.02, rec["f32"]         # fp32 agrees far closer than north_star's 0.1
    record("map_vs_oracle_one_interval", rec)


def test_two_fp16_pieces_against_exact_fp32_products_at_1000x600(world):
    """"two fp16 pieces == fp32", END TO END and against EXACT products.  The key graph bound with pieces = 0 runs every
    convolution on v_mfma_f32_32x32x2_f32 (each product an fp32 product, hip._conv_exact; FlowNet is not part of a first frame).  Frame 0
    at 1000x600 through both evaluations on the GPU: (i) the dense maps agree to fp32 round-off directly, (ii) with the exact evaluation in
    the fp32 oracle's place, the float64-anchored criterion (oracle/e2e.py) holds - the two-piece form is no further from float64 than
    1.5 x what exact fp32 products in an MFMA summation order are, and every proposal the two order differently is a float64 tie."""
    from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
    cfg, arg, aux, clip = (world[k] for k in ('cfg', 'arg', 'aux', 'clip'))
    if 'frame0' not in world:
        pytest.skip("needs test_first_key_cur_second_key_at_1000x600's frame 0")
    f0w = world['frame0']
    im_info = clip.im_info()
    im_t = torch.from_numpy(im_info).to(DEV)
    f0 = clip.frame(0).to(DEV)
    keyx = resnet_v1_101_flownet_rfcn(cfg).get_key_test_symbol(cfg).bind(arg, aux, DEV, pieces=0)
    keyx.taps = {}
    outx = keyx.forward(data=f0, im_info=im_t, data_key_old=f0, feat_key_old=torch.zeros(1, 1024, 1, 1, device=DEV))
    tx = dict(keyx.taps)
    keyx.check_status()
    rec = {}
    for name in ('backbone_feat', 'cls_map', 'box_map', 'rpn_bbox_pred'):
        two, ex, d64 = np_(f0w['taps'][name]), np_(tx[name]), f0w['d64'][name]
        rec[name] = dict(two_vs_exact=rel_err(two, ex), two_vs_f64=rel_err(two, d64), exact_vs_f64=rel_err(ex, d64), oracle_vs_f64=rel_err(f0w['ref'][name], d64))
        assert rec[name]['two_vs_exact'] < 1e-5, (name, rec[name])
        assert rec[name]['two_vs_f64'] <= e2e.RATIO * rec[name]['exact_vs_f64'] + 2.0 ** -23, (name, rec[name])
    rec['rpn_cls_prob_abs'] = float(np.abs(np_(f0w['taps']['rpn_cls_prob']) - np_(tx['rpn_cls_prob'])).max())
    gap = e2e.frame_gap(cfg, e2e.gpu_side(cfg, f0w['taps'], f0w['out'], im_info), e2e.gpu_side(cfg, tx, outx, im_info), f0w['d64'], im_info, H, W)
    rec['end_to_end_with_exact_fp32_as_the_yardstick'] = gap
    record('two_pieces_vs_exact_fp32', rec)
    assert not gap['failures'], gap['failures']
    assert gap['max_abs_dscore'] <= TOL_SCORE and gap['max_abs_dbox'] <= TOL_BOX_PX_BACKSTOP, gap


def trained_like_batchnorm(arg, aux, seed=5, octaves=8.0, outlier_share=0.01, outlier_log2=14.0):
    """The backbone's parameters with the per-channel spread a trained pre-activation ResNet has, instead of init_params' gamma = 1, beta = 0,
    mean = 0, var = 1 (dff_rfcn/core/callback.py:54-64, symbols/sym_common.py:92-102, resnet.py:70-101).  In a trained network the channels
    of a stage's residual stream differ in magnitude by many octaves and every bn1 that reads the stream undoes it: here channel c of stage
    s's stream is scaled by f_s[c] = 2^U(-octaves, octaves), `outlier_share` of the channels another 2^outlier_log2 up - rows c of every
    conv3 (and of the stage's shortcut) times f_s[c] - and the bn1 of every reader (the stage's later units, the next stage's first unit,
    the tail) gets gamma / sqrt(var) = 1 / f[c], a moving mean != 0 and the beta that makes up for it.  In real arithmetic the network
    computes what it computed before, so its detections stay as well-conditioned as the default parameters'; in fp32 the stored sums, the
    conv3 / shortcut weights (2^22 between their largest and smallest output channel: far more than the 2^15 window of one power-of-two
    scale per tensor) and the bn1 tables carry the spread."""
    from lsfa_amd.symbols import params as P
    rs = np.random.RandomState(seed)
    arg, aux = {k: np.array(v, copy=True) for k, v in arg.items()}, {k: np.array(v, copy=True) for k, v in aux.items()}
    f_prev = None
    for si in range(1, 5):
        C = arg['stage%d_unit1_conv3_weight' % si].shape[0]
        f = np.exp2(rs.uniform(-octaves, octaves, C))
        f[rs.rand(C) < outlier_share] *= 2.0 ** outlier_log2
        for u in range(1, P.UNITS[si - 1] + 1):
            p = 'stage%d_unit%d_' % (si, u)
            arg[p + 'conv3_weight'] = (arg[p + 'conv3_weight'].astype(np.float64) * f[:, None, None, None]).astype(np.float32)
            if u == 1:
                arg[p + 'sc_weight'] = (arg[p + 'sc_weight'].astype(np.float64) * f[:, None, None, None]).astype(np.float32)
            g = f if u > 1 else f_prev              # what this unit's bn1 reads: the stage's stream, or the previous stage's
            if g is not None:
                _rescale_bn(arg, aux, p + 'bn1', g, rs)
        f_prev = f
    _rescale_bn(arg, aux, 'bn1', f_prev, rs)       # the tail
    return arg, aux


def _rescale_bn(arg, aux, name, f, rs):
    """BatchNorm `name` reads a map whose channel c is f[c] times what it was: scale 1 / f[c], the mean moved off zero, beta making up for it"""
    C = f.shape[0]
    var = aux[name + '_moving_var'].astype(np.float64)
    s = arg[name + '_gamma'].astype(np.float64) / np.sqrt(var + 2e-5) / f
    m = rs.normal(0, 0.3, C) * f                                        # a mean of the size of the channel's values
    aux[name + '_moving_mean'] = (aux[name + '_moving_mean'].astype(np.float64) * f + m).astype(np.float32)
    arg[name + '_gamma'] = (s * np.sqrt(var + 2e-5)).astype(np.float32)
    arg[name + '_beta'] = (arg[name + '_beta'].astype(np.float64) + m * s).astype(np.float32)


def trained_like_everywhere(arg, aux, seed=6, octaves=4.0, outlier_share=0.001, outlier_log2=12.0):
    """trained_like_batchnorm's backbone + the same idea in every other convolution net of an interval: channel c of a map is
    scaled by f[c] = 2^U(-octaves, octaves), `outlier_share` of the channels another 2^outlier_log2 up - the producer's output rows (and bias)
    times f[c] - and every reader divides its input channels by f[c]; ReLU / LeakyReLU commute with a positive factor, so in real arithmetic
    the networks compute what they computed.  Spread this way:
      * the small net's stage-1 stream (conv3 / shortcut rows; its later units' bn1 undo it) - whose LAST reader is fuse_reduce_add, a
        convolution with NO BatchNorm in front: the two fp16 pieces there see a raw map with a 2^20 channel spread under ONE power-of-two scale,
        the matching weights spread the other way (resnet_v1_101_flownet_rfcn.py:209-236);
      * Nq_conv1 -> Nq_conv2 (:94-101), ReLU in between, no BatchNorm;
      * FlowNet's single-reader encoder maps flow_conv1 -> conv2, conv3 -> conv3_1, conv4 -> conv4_1, conv5 -> conv5_1, conv6 -> conv6_1 (:153-169).
    The window this is sized for: a value within 2^-16 of its map's maximum keeps all 22 bits of the two-piece form."""
    arg, aux = trained_like_batchnorm(arg, aux)
    rs = np.random.RandomState(seed)

    def spread(C):
        f = np.exp2(rs.uniform(-octaves, octaves, C))
        f[rs.rand(C) < outlier_share] *= 2.0 ** outlier_log2
        if not (f > 2.0 ** octaves).any():
            f[rs.randint(C)] *= 2.0 ** outlier_log2          # at least one outlier per map, whatever the share rounds to
        return f

    def rows(name, f, bias=True):
        arg[name + '_weight'] = (arg[name + '_weight'].astype(np.float64) * f[:, None, None, None]).astype(np.float32)
        if bias and name + '_bias' in arg:
            arg[name + '_bias'] = (arg[name + '_bias'].astype(np.float64) * f).astype(np.float32)

    def cols(name, f):
        arg[name + '_weight'] = (arg[name + '_weight'].astype(np.float64) / f[None, :, None, None]).astype(np.float32)

    # the small net's stage 1 (the only stage that exists: need_part) and its reader without a BatchNorm
    f = spread(arg['small_net_stage1_unit1_conv3_weight'].shape[0])
    for u in (1, 2, 3):
        p_ = 'small_net_stage1_unit%d_' % u
        rows(p_ + 'conv3', f, bias=False)
        if u == 1:
            rows(p_ + 'sc', f, bias=False)
        else:
            _rescale_bn(arg, aux, p_ + 'bn1', f, rs)
    cols('fuse_reduce_add', f)
    g = spread(arg['Nq_conv1_weight'].shape[0])
    rows('Nq_conv1', g)
    cols('Nq_conv2', g)
    for prod, cons in (('flow_conv1', 'conv2'), ('conv3', 'conv3_1'), ('conv4', 'conv4_1'), ('conv5', 'conv5_1'), ('conv6', 'conv6_1')):
        h = spread(arg[prod + '_weight'].shape[0])
        rows(prod, h)
        cols(cons, h)
    return arg, aux


def test_trained_like_statistics_over_an_interval_at_1000x600(world):
    """r5's trained-like test ran frame 0 only (backbone + heads) and spread the backbone's BatchNorms only.  Here the small
    net (whose output feeds fuse_reduce_add WITHOUT a BatchNorm), the Nq net and FlowNet carry a per-channel spread too (2^+-4, 0.1 % of the
    channels - at least one per map - 2^12 above that: activation outliers 2^12 x the map's typical magnitude entering convolutions that are
    not bn-at-the-cut), and a whole interval's kinds of frames go through: frame 0 (first key frame), frame 1 (non-key: small net +
    fuse_reduce_add + MV warp + heads) and frame 10 (second key frame: FlowNet + flow warp + Nq aggregation), each under the float64-anchored
    criterion (oracle/e2e.py), the dense maps within TOL_DENSE of the fp32 oracle's and no further from float64 than 1.5 x the oracle, and no
    status bit raised."""
    from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
    cfg, clip = world['cfg'], world['clip']
    arg, aux = trained_like_everywhere(world['arg'], world['aux'])
    im_info = clip.im_info()
    im_t = torch.from_numpy(im_info).to(DEV)
    net = resnet_v1_101_flownet_rfcn(cfg)
    key = net.get_key_test_symbol(cfg).bind(arg, aux, DEV)
    cur = net.get_cur_test_symbol(cfg).bind(arg, aux, DEV)
    p = graph_ref.Params(arg, aux)
    f0, f1, f10 = clip.frame(0), clip.frame(1), clip.frame(10)
    zero = np.zeros((1, 1024, 1, 1), np.float32)
    rec = {}
    # frame 0
    key.taps = {}
    out0 = key.forward(data=f0.to(DEV), im_info=im_t, data_key_old=f0.to(DEV), feat_key_old=torch.zeros(1, 1024, 1, 1, device=DEV))
    taps0 = dict(key.taps)
    ref0 = graph_ref.key_forward(cfg, arg, aux, f0.numpy(), f0.numpy(), zero, im_info)
    d0 = graph_ref.key_forward(cfg, arg, aux, f0.numpy(), f0.numpy(), zero, im_info, dtype=F64)
    rec['frame0'] = e2e.frame_gap(cfg, e2e.gpu_side(cfg, taps0, out0, im_info), ref0, d0, im_info, H, W)
    feat0 = out0['choose_feat_output']
    # frame 1: the non-key path
    mv, res = clip.motion_vector(1, 0), clip.res_diff(1)
    cur.taps = {}
    out1 = cur.forward(data=f1.to(DEV), im_info=im_t, feat_key=feat0, motion_vector=mv.to(DEV), res_diff=res.to(DEV))
    taps1 = dict(cur.taps)
    small_ref, small_64 = graph_ref.small_net_feature(p, f1).numpy(), graph_ref.small_net_feature(graph_ref.Params(arg, aux, dtype=F64), f1).numpy()
    rec['small_feat'] = dict(vs_oracle=rel_err(np_(taps1['small_feat']), small_ref), gpu_vs_f64=rel_err(np_(taps1['small_feat']), small_64),
                             oracle_vs_f64=rel_err(small_ref, small_64))
    check_cur_frame(cfg, arg, taps1, out1, feat0, mv, res, im_info)
    ref1 = graph_ref.cur_forward(cfg, arg, aux, f1.numpy(), ref0['choose_feat_output'], mv.numpy(), res.numpy(), im_info)
    d1 = graph_ref.cur_forward(cfg, arg, aux, f1.numpy(), d0['choose_feat_output'], mv.numpy(), res.numpy(), im_info, dtype=F64)
    rec['frame1'] = e2e.frame_gap(cfg, e2e.gpu_side(cfg, taps1, out1, im_info), ref1, d1, im_info, H, W)
    # frame 10: the second key frame
    key.taps = {}
    out10 = key.forward(data=f10.to(DEV), im_info=im_t, data_key_old=f0.to(DEV), feat_key_old=feat0)
    taps10 = dict(key.taps)
    flow_ref, scale_ref = graph_ref.get_flownet(p, f10, f0)
    rec['flow_abs'], rec['flow_max'] = float(np.abs(np_(taps10['flow']) - flow_ref.numpy()).max()), float(flow_ref.abs().max())
    rec['scale_map'] = rel_err(np_(taps10['scale_map']), scale_ref.numpy())
    check_key_frame(cfg, taps10, out10, feat0, im_info)
    logits_ref = graph_ref.nq_logits(p, torch.from_numpy(np_(taps10['warp'])), taps10['backbone_feat'].cpu()).numpy()
    rec['nq_logits_abs'], rec['nq_logits_max'] = float(np.abs(np_(taps10['nq_logits']) - logits_ref).max()), float(np.abs(logits_ref).max())
    ref10 = graph_ref.key_forward(cfg, arg, aux, f10.numpy(), f0.numpy(), ref0['choose_feat_output'], im_info)
    d10 = graph_ref.key_forward(cfg, arg, aux, f10.numpy(), f0.numpy(), d0['choose_feat_output'], im_info, dtype=F64)
    rec['frame10'] = e2e.frame_gap(cfg, e2e.gpu_side(cfg, taps10, out10, im_info), ref10, d10, im_info, H, W)
    rec['choose_feat_second_key_vs_f64'] = dict(gpu=rel_err(np_(out10['choose_feat_output']), d10['choose_feat_output']),
                                                oracle_fp32=rel_err(ref10['choose_feat_output'], d10['choose_feat_output']))
    key.check_status()
    cur.check_status()
    key.taps = cur.taps = None
    record('trained_like_everywhere', rec)
    assert rec['small_feat']['vs_oracle'] < TOL_DENSE, rec['small_feat']
    assert rec['small_feat']['gpu_vs_f64'] <= e2e.RATIO * rec['small_feat']['oracle_vs_f64'] + 2.0 ** -23, rec['small_feat']
    assert rec['scale_map'] < TOL_DENSE and rec['flow_abs'] < TOL_DENSE * max(1.0, rec['flow_max']), rec
    assert rec['nq_logits_abs'] < TOL_DENSE * max(1.0, rec['nq_logits_max']), rec
    for k in ('frame0', 'frame1', 'frame10'):
        e = rec[k]
        assert not e['failures'], (k, e['failures'], e)
        assert e['max_abs_dscore'] <= TOL_SCORE and e['max_abs_dbox'] <= TOL_BOX_PX_BACKSTOP, (k, e)
        assert e['rois_compared'] >= 250, (k, e)
    assert rec['choose_feat_second_key_vs_f64']['gpu'] <= e2e.RATIO * rec['choose_feat_second_key_vs_f64']['oracle_fp32'] + 2.0 ** -23, rec['choose_feat_second_key_vs_f64']


def test_trained_like_batchnorm_statistics_at_1000x600(world):
    """the fp32 path's two fp16 pieces take one power-of-two scale per activation MAP and - since r5 - one per OUTPUT
    CHANNEL of a weight.  With the residual streams, conv3 / shortcut weights and bn1 statistics spread the way a trained network's are
    (trained_like_batchnorm: 2^+-8 per channel, 1 % of the channels 2^14 above that, non-zero means) frame 0 at 1000x600 must still meet the float64-anchored criterion (oracle/e2e.py), the
    backbone feature must be no further from float64 than 1.5 x the fp32 oracle's, and no convolution may raise the overflow status."""
    from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
    cfg, clip = world['cfg'], world['clip']
    arg, aux = trained_like_batchnorm(world['arg'], world['aux'])
    im_info = clip.im_info()
    f0 = clip.frame(0)
    key = resnet_v1_101_flownet_rfcn(cfg).get_key_test_symbol(cfg).bind(arg, aux, DEV)
    # the channel spread the weights really have after folding (the premise of the test)
    spread = max(float(torch.log2(u['w3'].w_scale.max() / u['w3'].w_scale.min())) for u in key.net.units)
    assert spread >= 20.0, spread
    key.taps = {}
    out = key.forward(data=f0.to(DEV), im_info=torch.from_numpy(im_info).to(DEV), data_key_old=f0.to(DEV), feat_key_old=torch.zeros(1, 1024, 1, 1, device=DEV))
    taps = dict(key.taps)
    key.check_status()
    zero = np.zeros((1, 1024, 1, 1), np.float32)
    ref = graph_ref.key_forward(cfg, arg, aux, f0.numpy(), f0.numpy(), zero, im_info)
    d64 = graph_ref.key_forward(cfg, arg, aux, f0.numpy(), f0.numpy(), zero, im_info, dtype=F64)
    rec = {'weight_channel_spread_log2': spread,
           'backbone_feat_vs_f64': dict(gpu=rel_err(np_(taps['backbone_feat']), d64['backbone_feat']), oracle_fp32=rel_err(ref['backbone_feat'], d64['backbone_feat'])),
           'backbone_feat_vs_oracle': rel_err(np_(taps['backbone_feat']), ref['backbone_feat'])}
    gap = e2e.frame_gap(cfg, e2e.gpu_side(cfg, taps, out, im_info), ref, d64, im_info, H, W)
    rec['end_to_end'] = gap
    record('trained_like_batchnorm', rec)
    assert rec['backbone_feat_vs_oracle'] < TOL_DENSE, rec
    assert rec['backbone_feat_vs_f64']['gpu'] <= e2e.RATIO * rec['backbone_feat_vs_f64']['oracle_fp32'] + 2.0 ** -23, rec['backbone_feat_vs_f64']
    assert not gap['failures'], gap['failures']
    assert gap['max_abs_dscore'] <= TOL_SCORE and gap['max_abs_dbox'] <= TOL_BOX_PX_BACKSTOP, gap
