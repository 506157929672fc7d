"""BASELINE.json configs[0] ("single-frame R-FCN ResNet-101 on CPU, plumbing, no GPU"): the oracle's
statement of the key-frame graph on a first frame IS the single-frame R-FCN path
(backbone -> RPN -> Proposal -> PSROI -> decode -> NMS).  Run it on the CPU at a reduced size and
check the plumbing invariants the GPU path is later compared against."""
import numpy as np

import oracle
from oracle import graph_ref, np_ref
from lsfa_amd.config.config import lsfa_test_config
from lsfa_amd.symbols import params as P
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
from lsfa_amd.utils.synthetic import SyntheticClip


def test_single_frame_rfcn_graph_on_cpu_oracle():
    cfg = lsfa_test_config(10)
    cfg.network.add_dcn = False          # keeps the CPU run short; DCN is covered by the GPU graph test
    arg, aux = P.init_params(cfg, seed=5)
    clip = SyntheticClip(0, 1, 96, 160)
    im_info = clip.im_info()
    out = graph_ref.key_forward(cfg, arg, aux, clip.frame(0).numpy(), clip.frame(0).numpy(),
                                np.zeros((1, 1024, 1, 1), np.float32), im_info)
    assert out['choose_feat_output'].shape == (1, 1024, 6, 10)
    rois, prob, bbox = out['rois_output'], out['cls_prob_reshape_output'][0], out['bbox_pred_reshape_output'][0]
    assert rois.shape == (300, 5) and prob.shape == (300, 31) and bbox.shape == (300, 8)
    assert (rois[:, 0] == 0).all() and (rois[:, 1] >= 0).all() and (rois[:, 3] <= 159).all() and (rois[:, 4] <= 95).all()
    np.testing.assert_allclose(prob.sum(1), 1.0, atol=1e-5)
    # 6*10*9 = 540 anchors < 300 survivors is impossible after NMS -> the cyclic pad ran
    assert len(np.unique(rois, axis=0)) < 300
    dets, counts, _ = oracle.det_postprocess(rois, bbox, prob, 96, 160, 1.0)
    assert 0 < counts.sum() <= 300 + 30
    # the product's symbol object agrees with the oracle on names and shapes
    net = resnet_v1_101_flownet_rfcn(cfg)
    sym = net.get_key_test_symbol(cfg)
    net.infer_shape({'data': (1, 3, 96, 160), 'im_info': (1, 3)})
    assert net.out_shape_dict['rois_output'] == (300, 5)
    assert net.out_shape_dict['choose_feat_output'] == (1, 1024, 6, 10)
    net.check_parameter_shapes(arg, aux, {'data': (1, 3, 96, 160)})
    assert set(sym.arg_spec) <= set(arg) and set(sym.aux_spec) <= set(aux)
    assert 'stage3_unit23_conv2_offset_weight' not in sym.arg_spec     # add_dcn False
    cfg2 = lsfa_test_config(10)
    assert 'stage3_unit23_conv2_offset_weight' in resnet_v1_101_flownet_rfcn(cfg2).get_key_test_symbol(cfg2).arg_spec
    assert np_ref.key_frame_flags([1], 10) == [0]


def test_key_graph_aggregation_switches():
    """The three aggregation branches of the key symbol (symbols/resnet_v1_101_flownet_rfcn.py:310-315)
    in the oracle graph: the Fgfa weights are a 2-way softmax of cosine similarities, so the output lies
    between the warped and the current feature; the plain mean is exactly 0.5*(a+b)."""
    import torch
    outs = {}
    for mode in ("nq", "fgfa", "average"):
        cfg = lsfa_test_config(key_frame_interval=10)
        cfg.network.add_Nq_net = (mode == "nq")
        cfg.network.add_Fgfa_net = (mode == "fgfa")
        arg, aux = P.init_params(cfg, seed=5)
        clip = SyntheticClip(1, 12, 64, 96)
        f0, f10 = clip.frame(0).numpy(), clip.frame(10).numpy()
        im_info = clip.im_info()
        r0 = graph_ref.key_forward(cfg, arg, aux, f0, f0, np.zeros((1, 1024, 1, 1), np.float32), im_info)
        r = graph_ref.key_forward(cfg, arg, aux, f10, f0, r0['choose_feat_output'], im_info)
        lo = np.minimum(r['warp'], r['backbone_feat'])
        hi = np.maximum(r['warp'], r['backbone_feat'])
        tol = 1e-5 * np.maximum(np.abs(lo), np.abs(hi)) + 1e-12
        assert (r['choose_feat_output'] >= lo - tol).all() and (r['choose_feat_output'] <= hi + tol).all()
        outs[mode] = r
    mean = 0.5 * (torch.from_numpy(outs['average']['warp']) + torch.from_numpy(outs['average']['backbone_feat']))
    np.testing.assert_array_equal(outs['average']['choose_feat_output'], mean.numpy())
    assert 'embed' in outs['fgfa'] and outs['fgfa']['embed'].shape[:2] == (2, 2048)
    assert 'nq_logits' in outs['nq']


def _side(out):
    return {k: out[k] for k in ('rpn_cls_prob', 'rpn_bbox_pred', 'rois_output', 'roi_anchor', 'cls_prob_reshape_output',
                                'bbox_pred_reshape_output')}


def test_float64_anchored_criterion_accepts_fp32_and_rejects_a_degraded_side():
    """oracle/e2e.py on the CPU: the fp32 oracle graph judged against the float64 graph passes with itself as the 'GPU'
    side (ratios 1), still passes when the 'GPU' side is another fp32 evaluation (rounded through float64 -> fp32 maps),
    and FAILS when the RPN scores of the 'GPU' side carry errors 10x the fp32 oracle's (what a dropped partial product
    in a convolution would do), or when two proposals that are not a float64 tie are swapped."""
    import torch
    from oracle import e2e
    cfg = lsfa_test_config(key_frame_interval=10)
    arg, aux = P.init_params(cfg, seed=0)
    H, W = 192, 320
    clip = SyntheticClip(0, 12, H, W)
    f0, im_info = clip.frame(0).numpy(), clip.im_info()
    z = np.zeros((1, 1024, 1, 1), np.float32)
    r32 = graph_ref.key_forward(cfg, arg, aux, f0, f0, z, im_info)
    r64 = graph_ref.key_forward(cfg, arg, aux, f0, f0, z, im_info, dtype=torch.float64)
    assert r64['backbone_feat'].dtype == np.float64
    # the fp32 statement is an fp32-accurate evaluation of the float64 one
    assert np.abs(r32['backbone_feat'] - r64['backbone_feat']).max() < 1e-5 * np.abs(r64['backbone_feat']).max()
    rec = e2e.frame_gap(cfg, _side(r32), _side(r32), _side(r64), im_info, H, W)
    assert not rec['failures'], rec
    assert rec['roi_order_identical'] and rec['survivor_mismatch'] == 0
    assert rec['err_vs_f64_rpn_score']['ratio'] == 1.0

    # a degraded side: scores off by 10x the oracle's own error -> (a) fails
    bad = dict(_side(r32))
    e_ref = rec['err_vs_f64_rpn_score']['oracle_fp32']
    rng = np.random.RandomState(0)
    bad['rpn_cls_prob'] = (r32['rpn_cls_prob'] + rng.uniform(-1, 1, r32['rpn_cls_prob'].shape) * (10 * e_ref + 1e-6)).astype(np.float32)
    rec_bad = e2e.frame_gap(cfg, bad, _side(r32), _side(r64), im_info, H, W)
    assert any(f.startswith('rpn_score') for f in rec_bad['failures']), rec_bad['failures']

    # two proposals swapped that are no tie in float64 -> (b) fails
    sw = dict(_side(r32))
    a = np.array(r32['roi_anchor'])
    s64 = e2e.fg_scores(r64['rpn_cls_prob'], cfg.network.NUM_ANCHORS)
    i = next(i for i in range(len(a) - 1) if abs(s64[a[i]] - s64[a[i + 1]]) > 1e-3)
    for k in ('rois_output', 'roi_anchor'):
        v = np.array(r32[k]); v[[i, i + 1]] = v[[i + 1, i]]; sw[k] = v
    for k in ('cls_prob_reshape_output', 'bbox_pred_reshape_output'):
        v = np.array(r32[k]); v[0, [i, i + 1]] = v[0, [i + 1, i]]; sw[k] = v
    rec_sw = e2e.frame_gap(cfg, sw, _side(r32), _side(r64), im_info, H, W)
    assert rec_sw['roi_displaced'] == 2 and any('ordered differently' in f for f in rec_sw['failures']), rec_sw['failures']
