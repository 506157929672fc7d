"""LSFA test graphs (key-frame and non-key-frame) on MI355X.

API mirror of dff_rfcn/symbols/resnet_v1_101_flownet_rfcn.py: the class
`resnet_v1_101_flownet_rfcn(cfg)` with `get_key_test_symbol(cfg)` (:448-551),
`get_cur_test_symbol(cfg)` (:553-659) and `init_weight(cfg, arg_params, aux_params)` (:753-870).
A "symbol" here is a `TestSymbol`: it knows its argument / auxiliary / output names and
shapes like an mx.sym.Group, and `bind()`s to an executor that runs the graph.

Execution design (not a translation of the MXNet graph):
  * dense contractions (ResNet-101, FlowNet, small net, Nq convs, 1x1 heads) go to MIOpen /
    hipBLASLt through PyTorch-ROCm; inference BatchNorms are folded into the preceding
    convolution at bind time where the graph allows it (bn2/bn3 of each pre-activation unit,
    bn0), the remaining ones (bn1 of each unit, bn_data, final bn1) run as one fused
    scale-shift-ReLU HIP pass; 1x1 convolutions run as GEMMs with the residual add or bias fused
    into the GEMM epilogue; rpn_inv_normalize (operator_py/rpn_inv_normalize.py:19-26) is
    folded into rpn_bbox_pred's weights;
  * warp, x scale_map, + rnet_conv0(res_diff), + small-net feature, the Nq softmax-combine,
    Proposal, PSROI pooling + 7x7 average + class softmax, and DCN's bilinear im2col are the
    hand-written HIP kernels behind include/lsfa_hip.h;
  * ChooseOldKeyFeat / ChooseFeat (operator_py/choose_old_key_feat.py:23-32, choose_feat.py:23-31)
    are a host-side `if` on the SHAPE of feat_key_old, exactly the reference's first-frame test,
    without its GPU->CPU sync; on the first frame FlowNet / warp / Nq are skipped because
    ChooseFeat discards their result.
"""
import numpy as np
import torch
import torch.nn.functional as F

from lsfa_amd import hip
from lsfa_amd.symbols import params as P

BN_EPS = 2e-5  # sym_common.py:9
import os as _os
# measured at 1000x600 fp32 (us, hipGraph replay): backbone 5806 -> 5476 (5175 with the channels-last DCN im2col),
# small net 270 -> 258.  FlowNet stays NCHW: its large-kernel strided convolutions and deconvolutions were
# slower channels-last (1164 -> 1566 us).
_CL_DEFAULT = 'backbone,small'
_FLOW_GEMM_MAX_L = int(_os.environ.get('LSFA_FLOW_GEMM_L', '700'))   # FlowNet convs with at most this many output pixels run as im2col + GEMM
_CONV1X1_MIOPEN = _os.environ.get('LSFA_CONV1X1_MIOPEN', '0') == '1'   # experiment: channel-reducing 1x1 convs through MIOpen
# which 3x3 convolutions of the channels-last sub-networks run on the own fp32-MFMA implicit GEMM (lsfa_conv_nhwc_fwd,
# bias + ReLU in its epilogue) instead of the library's kernel + a separate bias/ReLU pass: comma list of backbone, small,
# fuse (the small net's 256 -> 1024 fuse_reduce_add convolution) and feat (feat_conv_3x3), both on the split-bf16 kernel;
# stem (bn_data + conv0 + relu0 + pool0 of both ResNets and the 4x4 average pooling in front of the small net: stem.hip);
# `conv3` additionally runs the 1x1 conv3 of those units on it with the shortcut add and the next unit's bn1 + ReLU fused
# `pw` (r3): EVERY contraction of a unit on the split-bf16 kernel — conv1, the shortcut, conv3 with the shortcut add and the
# next unit's bn1 + ReLU in its epilogue, the DCN offset branch and contraction — so that no library GEMM / convolution and no
# separate BatchNorm pass is left in the ResNets (and results no longer depend on a library's per-process algorithm choice);
# (lsfa_conv_nhwc_fused_fwd) - measured SLOWER (backbone 4239 -> 4983 us: the 64x64-tile kernel loses to the tuned library GEMM on
# K = 256 by more than the saved BN pass), so it is off by default.
# measured at 1000x600 (tools/key_sections.py, hipGraph replay): backbone 4327 -> 4239 us with it, small net 256 -> 289 us:
# the backbone's stage 2/3 units gain (epilogue fusion + deterministic tap split), the small net's 64-channel stage 1 and the
# 256 -> 1024 fuse convolution do not
_OWN_CONV = set(x for x in _os.environ.get('LSFA_OWN_CONV', 'backbone,small,fuse,feat,stem,flow,nq,h3').split(',') if x)
# the own 3x3 convolutions on the bf16 matrix pipe with exactly split fp32 operands (lsfa_conv_split_fwd: fp32 in, fp32
# accumulate, error against float64 equal to the fp32-MFMA kernel's) instead of the fp32 matrix instructions.  Measured per
# conv2 at 1000x600 (tools/lab/conv_split_lab.py): res4 32.7 vs 40.7 us, res3 31.4 vs 50.4, res2 34.6 vs 51.3.
_CONV_SPLIT = _os.environ.get('LSFA_CONV_SPLIT', '1') == '1'
_UNIT_TAPS = _os.environ.get('LSFA_UNIT_TAPS', '0') == '1'       # diagnostics: every unit's a / c1 / c2 / shortcut / output as taps


class TestSymbol(object):
    """What get_*_test_symbol returns: names + shapes + bind()."""

    def __init__(self, kind, cfg):
        assert kind in ('key', 'cur', 'batch')
        self.kind = kind
        self.cfg = cfg
        self.arg_spec, self.aux_spec = {'key': P.key_symbol_spec, 'cur': P.cur_symbol_spec,
                                        'batch': P.batch_symbol_spec}[kind](cfg)
        if kind == 'batch':
            self.data_names = ['data_key', 'data_other', 'im_info']
        else:
            self.data_names = ['data', 'im_info', 'data_key', 'data_key_old', 'motion_vector', 'res_diff',
                               'feat_key_old', 'feat_key']

    def list_arguments(self):
        return list(self.data_names) + list(self.arg_spec.keys())

    def list_auxiliary_states(self):
        return list(self.aux_spec.keys())

    def list_outputs(self):
        if self.kind == 'batch':  # Group at :749
            return ['rois_output', 'cls_prob_reshape_output', 'bbox_pred_reshape_output']
        if self.kind == 'key':   # Group at :549
            return ['data_key', 'motion_vector', 'res_diff', 'feat_key', 'choose_feat_output', 'rois_output',
                    'cls_prob_reshape_output', 'bbox_pred_reshape_output']
        return ['data', 'data_key', 'data_key_old', 'feat_key_old', 'rois_output', 'cls_prob_reshape_output',
                'bbox_pred_reshape_output']   # Group at :657

    def infer_shape(self, **data_shapes):
        if self.kind == 'batch':
            nb = 1 + data_shapes['data_other'][0]
            post, ncls = self.cfg.TEST.RPN_POST_NMS_TOP_N, self.cfg.dataset.NUM_CLASSES
            nreg = 2 if self.cfg.CLASS_AGNOSTIC else ncls
            outs = [(nb * post, 5), (self.cfg.TEST.BATCH_IMAGES, nb * post // self.cfg.TEST.BATCH_IMAGES, ncls),
                    (self.cfg.TEST.BATCH_IMAGES, nb * post // self.cfg.TEST.BATCH_IMAGES, 4 * nreg)]
            args = [tuple(data_shapes.get(k)) if k in data_shapes else None for k in self.data_names] + list(self.arg_spec.values())
            return args, outs, list(self.aux_spec.values())
        n, _, h, w = data_shapes['data']
        fh, fw = int(np.ceil(h / 16.0)), int(np.ceil(w / 16.0))
        post = self.cfg.TEST.RPN_POST_NMS_TOP_N
        ncls = self.cfg.dataset.NUM_CLASSES
        nreg = 2 if self.cfg.CLASS_AGNOSTIC else ncls
        out = {'rois_output': (post, 5), 'cls_prob_reshape_output': (self.cfg.TEST.BATCH_IMAGES, post, ncls),
               'bbox_pred_reshape_output': (self.cfg.TEST.BATCH_IMAGES, post, 4 * nreg),
               'choose_feat_output': (n, self.cfg.network.DFF_FEAT_DIM, fh, fw)}
        for k in self.list_outputs():
            if k in data_shapes:
                out[k] = tuple(data_shapes[k])
        arg_shapes = [tuple(data_shapes[k]) if k in data_shapes else None for k in self.data_names] + \
                     list(self.arg_spec.values())
        return arg_shapes, [out.get(k) for k in self.list_outputs()], list(self.aux_spec.values())

    def bind(self, arg_params, aux_params, device='cuda:0', dtype=torch.float32):
        return Executor(self, arg_params, aux_params, device, dtype)


class resnet_v1_101_flownet_rfcn(object):
    def __init__(self, cfg):
        if cfg.network.nettype != 'resnet' or cfg.network.num_layer != 101:
            raise RuntimeError("unknow nettype: %s" % cfg.network.nettype)
        self.cfg = cfg
        self.sym = None
        self.arg_shape_dict = self.out_shape_dict = self.aux_shape_dict = None

    @property
    def symbol(self):
        return self.sym

    def get_key_test_symbol(self, cfg):
        self.sym = TestSymbol('key', cfg)
        return self.sym

    def get_cur_test_symbol(self, cfg):
        self.sym = TestSymbol('cur', cfg)
        return self.sym

    def get_batch_test_symbol(self, cfg):
        self.sym = TestSymbol('batch', cfg)
        return self.sym

    def get_train_symbol(self, cfg):
        raise NotImplementedError("training is out of scope (SURVEY.md §8)")

    # lib/utils/symbol.py:36-55
    def infer_shape(self, data_shape_dict):
        arg_shape, out_shape, aux_shape = self.sym.infer_shape(**data_shape_dict)
        self.arg_shape_dict = dict(zip(self.sym.list_arguments(), arg_shape))
        self.out_shape_dict = dict(zip(self.sym.list_outputs(), out_shape))
        self.aux_shape_dict = dict(zip(self.sym.list_auxiliary_states(), aux_shape))

    def check_parameter_shapes(self, arg_params, aux_params, data_shape_dict, is_train=False):
        for k in self.sym.list_arguments():
            if k in data_shape_dict or k in self.sym.data_names:
                continue
            assert k in arg_params, k + ' not initialized'
            assert tuple(arg_params[k].shape) == tuple(self.arg_shape_dict[k]), \
                'shape inconsistent for ' + k + ' inferred ' + str(self.arg_shape_dict[k]) + ' provided ' + str(
                    tuple(arg_params[k].shape))
        for k in self.sym.list_auxiliary_states():
            assert k in aux_params, k + ' not initialized'
            assert tuple(aux_params[k].shape) == tuple(self.aux_shape_dict[k]), \
                'shape inconsistent for ' + k

    def init_weight(self, cfg, arg_params, aux_params, seed=0):
        """Fill in whatever the loaded checkpoint lacks, like :753-870: small_net_* copied from the
        big net, normal(0, 0.01) / zeros for the layers LSFA adds."""
        rnd_arg, rnd_aux = P.init_params(cfg, seed)
        for k in self.sym.arg_spec:
            if k not in arg_params:
                src = k.replace('small_net_', '')
                arg_params[k] = arg_params[src].copy() if ('small_net_' in k and src in arg_params) else rnd_arg[k]
        for k in self.sym.aux_spec:
            if k not in aux_params:
                src = k.replace('small_net_', '')
                aux_params[k] = aux_params[src].copy() if ('small_net_' in k and src in aux_params) else rnd_aux[k]


# ======================================================================================
def _t(a, device, dtype=torch.float32):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dtype)


class _ResNetWeights(object):
    """Folded weights of a (prefix) pre-activation ResNet-101 (or its stem + stage 1)."""

    def __init__(self, arg, aux, prefix, stages, add_dcn, tail, device, cdtype):
        f32 = torch.float32

        def bn(name):
            g = arg[name + '_gamma'].astype(np.float64)
            if name.endswith('bn_data'):
                g = np.ones_like(g)            # fix_gamma=True, resnet.py:151
            b = arg[name + '_beta'].astype(np.float64)
            m = aux[name + '_moving_mean'].astype(np.float64)
            v = aux[name + '_moving_var'].astype(np.float64)
            s = g / np.sqrt(v + BN_EPS)
            return s, b - m * s

        def fold(w, s):
            return (w.astype(np.float64) * s.reshape(-1, 1, 1, 1)).astype(np.float32)

        s, t = bn(prefix + 'bn_data')
        self.bn_data = (_t(s.astype(np.float32), device, f32), _t(t.astype(np.float32), device, f32))
        s0, t0 = bn(prefix + 'bn0')
        self.conv0_w = _t(fold(arg[prefix + 'conv0_weight'], s0), device, cdtype)
        self.conv0_b = _t(t0.astype(np.float32), device, cdtype)
        self.units = []
        for si in range(1, stages + 1):
            for u in range(1, P.UNITS[si - 1] + 1):
                p = '%sstage%d_unit%d_' % (prefix, si, u)
                s1, t1 = bn(p + 'bn1')
                s2, t2 = bn(p + 'bn2')
                s3, t3 = bn(p + 'bn3')
                d = dict(stage=si, unit=u,
                         bn1=(_t(s1.astype(np.float32), device, f32), _t(t1.astype(np.float32), device, f32)),
                         w1=_t(fold(arg[p + 'conv1_weight'], s2), device, cdtype), b1=_t(t2.astype(np.float32), device, cdtype),
                         w2=_t(fold(arg[p + 'conv2_weight'], s3), device, cdtype), b2=_t(t3.astype(np.float32), device, cdtype),
                         w3=_t(arg[p + 'conv3_weight'], device, cdtype), dcn=P.is_dcn_unit(si, u, add_dcn))
                if u == 1:
                    d['sc'] = _t(arg[p + 'sc_weight'], device, cdtype)
                if d['dcn']:
                    d['off_w'] = _t(arg[p + 'conv2_offset_weight'], device, cdtype)
                    d['off_b'] = _t(arg[p + 'conv2_offset_bias'], device, cdtype)
                    d['w2_2d'] = d['w2'].reshape(d['w2'].shape[0], -1).contiguous()
                d['w1_2d'] = d['w1'].reshape(d['w1'].shape[0], -1)
                d['w3_2d'] = d['w3'].reshape(d['w3'].shape[0], -1)
                self.units.append(d)
        if tail:
            s, t = bn(prefix + 'bn1')
            self.bn1 = (_t(s.astype(np.float32), device, f32), _t(t.astype(np.float32), device, f32))
        self._cl_ready = False

    def prepare_channels_last(self):
        """Weight forms of the channels-last path: (Cin, Cout) matrices for the 1x1 convolutions (GEMM on
        (H*W, C) rows), channels_last 4-D weights for the library convolutions."""
        if self._cl_ready:
            return
        cl = torch.channels_last
        self.conv0_w_cl = self.conv0_w.contiguous(memory_format=cl)
        for d in self.units:
            d['w1_t'] = d['w1_2d'].t().contiguous()
            d['w3_t'] = d['w3_2d'].t().contiguous()
            d['w2_cl'] = d['w2'].contiguous(memory_format=cl)
            if d['w2'].dtype == torch.float32 and d['w2'].shape[1] % 32 == 0 and d['w2'].shape[0] % 64 == 0:
                d['w2_kc'] = hip.conv_weight_kc(d['w2'])        # (Cout, 9, Cin): lsfa_conv_nhwc_fwd's layout
                if _CONV_SPLIT and d['w2'].is_cuda:
                    d['w2_split'] = hip.SplitWeight(d['w2'])    # three bf16 pieces per weight, MFMA fragment order
            if d['w3'].dtype == torch.float32 and d['w3'].shape[1] % 32 == 0 and d['w3'].shape[0] % 64 == 0:
                d['w3_kc'] = hip.conv_weight_kc(d['w3'])        # (Cout, 1, Cin)
            if 'sc' in d:
                d['sc_t'] = d['sc'].reshape(d['sc'].shape[0], -1).t().contiguous()
                d['sc_cl'] = d['sc'].contiguous(memory_format=cl)
            if d['dcn']:
                d['off_w_cl'] = d['off_w'].contiguous(memory_format=cl)
                # (Cout, C, 3, 3) -> rows ordered (tap, c) to match lsfa_deform_im2col_cl's col
                d['w2_tap_t'] = d['w2'].permute(2, 3, 1, 0).reshape(-1, d['w2'].shape[0]).contiguous()
            if _CONV_SPLIT and d['w1'].is_cuda and d['w1'].dtype == torch.float32:
                # what would otherwise go to MIOpen (whose algorithm choice depends on per-user state): the
                # stride-2 1x1 shortcuts and the DCN units' offset branch + contraction
                if 'sc' in d and d['stage'] in (2, 3):
                    d['sc_split'] = hip.SplitWeight(d['sc'])
                if d['dcn']:
                    co = d['off_w'].shape[0]
                    pad_to = -(-co // 64) * 64
                    wpad = torch.zeros((pad_to,) + tuple(d['off_w'].shape[1:]), device=d['off_w'].device, dtype=torch.float32)
                    wpad[:co] = d['off_w']
                    bpad = torch.zeros(pad_to, device=d['off_w'].device, dtype=torch.float32)
                    bpad[:co] = d['off_b']
                    d['off_split'], d['off_b_pad'] = hip.SplitWeight(wpad, real_cout=co), bpad
                    d['dcn_split'] = hip.SplitWeight(d['w2_tap_t'].t().contiguous().view(d['w2'].shape[0], -1, 1, 1))
                    if 'h3' in _OWN_CONV and d['w2'].shape[0] % 128 == 0:
                        d['dcn_split_h'] = hip.SplitWeightH(d['w2_tap_t'].t().contiguous().view(d['w2'].shape[0], -1, 1, 1))
            if _CONV_SPLIT and 'pw' in _OWN_CONV and d['w1'].is_cuda and d['w1'].dtype == torch.float32:
                # every contraction of the unit on lsfa_conv_split_fwd: the 1x1 convolutions, the shortcut, and for a DCN
                # unit the offset branch (72 output channels, zero-padded to 128) and the contraction of the sampled
                # columns (a 1x1 convolution over 9*C "channels" ordered (tap, c))
                d['w1_split'] = hip.SplitWeight(d['w1'])
                d['w3_split'] = hip.SplitWeight(d['w3'])
                if 'sc' in d and 'sc_split' not in d:
                    d['sc_split'] = hip.SplitWeight(d['sc'])
            if _CONV_SPLIT and 'conv3s' in _OWN_CONV and 'w3_split' not in d and d['w3'].is_cuda and d['w3'].dtype == torch.float32:
                d['w3_split'] = hip.SplitWeight(d['w3'])      # conv3 alone on the split kernel (residual + next bn1/ReLU fused)
        self._cl_ready = True


class Executor(object):
    """A bound test symbol: folded weights on the device + forward()."""

    def __init__(self, sym, arg_params, aux_params, device, dtype):
        self.sym, self.cfg = sym, sym.cfg
        self.device = torch.device(device)
        self.cdtype = dtype           # dtype of the dense contractions (fp32, or bf16 for config 3)
        self.taps = None              # set to {} to record stage outputs (parity tests)
        self._const = {}
        # which sub-networks run channels-last (activations as (H*W, C) rows): LSFA_CL = comma list of
        # backbone, small; fp32 contractions only
        want = _os.environ.get('LSFA_CL', _CL_DEFAULT)
        self.cl = set(x for x in want.split(',') if x) if dtype == torch.float32 else set()
        cfg = self.cfg
        arg = {k: np.asarray(v, dtype=np.float32) for k, v in arg_params.items()}
        aux = {k: np.asarray(v, dtype=np.float32) for k, v in aux_params.items()}
        for k, shp in sym.arg_spec.items():
            if k not in arg:
                raise KeyError(k + ' not initialized')
            if tuple(arg[k].shape) != tuple(shp):
                raise ValueError('shape inconsistent for %s inferred %s provided %s' % (k, shp, arg[k].shape))
        dev, cd = self.device, self.cdtype
        A = cfg.network.NUM_ANCHORS
        # heads: one GEMM for both RPN convs, one for both R-FCN convs
        std = np.tile(np.asarray(cfg.network.ANCHOR_STDS, np.float32), A)
        mean = np.tile(np.asarray(cfg.network.ANCHOR_MEANS, np.float32), A)
        wb, bb = arg['rpn_bbox_pred_weight'].reshape(4 * A, 512), arg['rpn_bbox_pred_bias']
        if cfg.network.NORMALIZE_RPN:   # rpn_inv_normalize folded: (Wx+b)*std+mean
            wb, bb = wb * std[:, None], bb * std + mean
        self.rpn_w = _t(np.concatenate([arg['rpn_cls_score_weight'].reshape(2 * A, 512), wb], 0), dev, cd)
        self.rpn_b = _t(np.concatenate([arg['rpn_cls_score_bias'], bb], 0), dev, cd)
        self.rfcn_w = _t(np.concatenate([arg['rfcn_cls_weight'].reshape(-1, 512), arg['rfcn_bbox_weight'].reshape(-1, 512)], 0), dev, cd)
        self.rfcn_b = _t(np.concatenate([arg['rfcn_cls_bias'], arg['rfcn_bbox_bias']], 0), dev, cd)
        self.n_cls_ch = arg['rfcn_cls_weight'].shape[0]
        # position-sensitive layout: GEMM [HW,512] x [512, 49*(ncls+nbox)], row (bin*D + d) of the permuted weight
        G = 7
        self.ncls = self.n_cls_ch // (G * G)
        self.nbox = arg['rfcn_bbox_weight'].shape[0] // (G * G)
        wc = arg['rfcn_cls_weight'].reshape(self.ncls, G * G, 512)
        wx = arg['rfcn_bbox_weight'].reshape(self.nbox, G * G, 512)
        w_ps = np.concatenate([wc, wx], 0).transpose(1, 0, 2).reshape(-1, 512)          # (49*D, 512)
        b_ps = np.concatenate([arg['rfcn_cls_bias'].reshape(self.ncls, G * G), arg['rfcn_bbox_bias'].reshape(self.nbox, G * G)], 0).T.reshape(-1)
        self.rfcn_w_ps_t = _t(np.ascontiguousarray(w_ps.T), dev, cd)                  # (512, 49*D)
        self.rfcn_b_ps = _t(b_ps, dev, cd)
        self.ps_layout = True
        self.proposal = hip.ProposalOp(feature_stride=cfg.network.RPN_FEAT_STRIDE, scales=cfg.network.ANCHOR_SCALES,
                                       ratios=cfg.network.ANCHOR_RATIOS, rpn_pre_nms_top_n=cfg.TEST.RPN_PRE_NMS_TOP_N,
                                       rpn_post_nms_top_n=cfg.TEST.RPN_POST_NMS_TOP_N, threshold=cfg.TEST.RPN_NMS_THRESH,
                                       rpn_min_size=cfg.TEST.RPN_MIN_SIZE)
        if sym.kind in ('key', 'batch'):
            self.net = _ResNetWeights(arg, aux, '', 4, cfg.network.add_dcn, True, dev, cd)
            self.feat_w, self.feat_b = _t(arg['feat_conv_3x3_weight'], dev, cd), _t(arg['feat_conv_3x3_bias'], dev, cd)
            self.flow = {k: _t(v, dev, cd) for k, v in arg.items()
                         if k.startswith(('flow_conv1', 'conv', 'Convolution', 'deconv', 'upsample_flow')) and 'stage' not in k and not k.startswith('conv0')}
            if sym.kind == 'batch':
                pass
            elif cfg.network.add_Nq_net:
                self.nq = [(_t(arg['Nq_conv%d_weight' % i], dev, cd), _t(arg['Nq_conv%d_bias' % i], dev, cd)) for i in (1, 2, 3)]
            elif cfg.network.add_Fgfa_net:
                self.em = [(_t(arg['em_conv%d_weight' % i], dev, cd), _t(arg['em_conv%d_bias' % i], dev, cd)) for i in (1, 2, 3)]
        else:
            self.rnet_w = _t(arg['rnet_conv0_weight'].reshape(1024, 3), dev, torch.float32)
            self.rnet_b = _t(arg['rnet_conv0_bias'], dev, torch.float32)
            if cfg.network.add_small_net:
                self.small = _ResNetWeights(arg, aux, 'small_net_', 1, False, False, dev, cd)
                self.fuse_w, self.fuse_b = _t(arg['fuse_reduce_add_weight'], dev, cd), _t(arg['fuse_reduce_add_bias'], dev, cd)

    def _tap(self, name, x):
        if self.taps is not None:
            self.taps[name] = x

    # ---- dense helpers ---------------------------------------------------------------
    def _c(self, x):
        return x if x.dtype == self.cdtype else x.to(self.cdtype)

    def _conv1x1(self, x, w2d, bias=None, residual=None, relu=False):
        """1x1 stride-1 convolution as a GEMM on the NCHW map.  `residual` is accumulated IN PLACE
        (GEMM beta = 1 into the shortcut tensor, no copy); a bias is applied together with the ReLU by
        the fused scale-shift-ReLU pass instead of a broadcast-copy + GEMM + ReLU triple."""
        n, c, h, w = x.shape
        if n != 1:
            y = F.conv2d(x, w2d.view(w2d.shape[0], c, 1, 1), bias)
            y = y if residual is None else y.add_(residual)
            return torch.relu_(y) if relu else y
        X = x.view(c, h * w)
        if residual is not None:
            out = residual.view(-1, h * w).addmm_(w2d, X)
        elif _CONV1X1_MIOPEN and w2d.shape[0] < c:
            out = F.conv2d(x, w2d.view(w2d.shape[0], c, 1, 1))
        else:
            out = torch.mm(w2d, X)
        out = out.view(1, -1, h, w)
        if bias is not None or relu:
            out = self._bias_act(out, bias, relu)
        return out

    def _bias_act(self, y, bias, relu):
        """y = max(y + bias[c], 0) in one in-place pass (fp32 maps; other dtypes use torch)."""
        if y.dtype == torch.float32:
            c = y.shape[1]
            if bias is None:
                bias = self._zeros(c)
            return hip.scale_shift_relu(y, self._ones(c), bias, relu=relu, out=y)
        if bias is not None:
            y = y.add_(bias.view(1, -1, 1, 1))
        return torch.relu_(y) if relu else y

    def _ones(self, c):
        t = self._const.get(('1', c))
        if t is None:
            t = self._const[('1', c)] = torch.ones(c, device=self.device)
        return t

    def _zeros(self, c):
        t = self._const.get(('0', c))
        if t is None:
            t = self._const[('0', c)] = torch.zeros(c, device=self.device)
        return t

    def _dcn(self, x, u, dilate):
        """DeformableConvolution (sym_common.py:138-157): offsets by an ordinary conv, bilinear
        im2col in HIP, then the contraction as one GEMM with the folded bn3 bias."""
        off = F.conv2d(x, u['off_w'], u['off_b'], stride=1, padding=dilate, dilation=dilate)
        col = hip.deform_im2col(x.float(), off.float(), 3, 3, dilate, 1, dilate, P.NUM_DEFORMABLE_GROUP)
        n, _, h, w = off.shape
        out = torch.matmul(u['w2_2d'], self._c(col))       # (n, Cout, Ho*Wo); bn3's folded bias joins the ReLU pass
        return out.view(n, -1, h, w)

    def _resnet(self, x, net, stages, tail):
        """Pre-activation ResNet (resnet.py:138-240); returns the stage outputs the caller asked for."""
        x = hip.scale_shift_relu(x, net.bn_data[0], net.bn_data[1], relu=False)
        x = self._bias_act(F.conv2d(self._c(x), net.conv0_w, None, stride=2, padding=3), net.conv0_b, True)
        x = F.max_pool2d(x, 3, 2, 1)
        dilate = 1
        for u in net.units:
            if u['stage'] > stages:
                break
            first = u['unit'] == 1
            # stage 4 keeps stride 1 and doubles the dilation from its 2nd unit on (resnet.py:33-34, :72-76, :223-230)
            stride = 2 if (first and u['stage'] in (2, 3)) else 1
            unit_dilate = dilate
            if first and u['stage'] == 4:
                dilate = dilate * 2
            a = self._c(hip.scale_shift_relu(x.float(), u['bn1'][0], u['bn1'][1], relu=True))
            c1 = self._conv1x1(a, u['w1_2d'], bias=u['b1'], relu=True)
            if u['dcn']:
                c2 = self._dcn(c1, u, unit_dilate)
            else:
                c2 = F.conv2d(c1, u['w2'], None, stride=stride, padding=unit_dilate, dilation=unit_dilate)
            c2 = self._bias_act(c2, u['b2'], True)
            if first:
                sc = F.conv2d(a, u['sc'], None, stride=stride) if stride != 1 else self._conv1x1(a, u['sc'].view(u['sc'].shape[0], -1))
            else:
                sc = x if x.dtype == self.cdtype else self._c(x)   # overwritten in place by conv3's GEMM (beta = 1)
            x = self._conv1x1(c2, u['w3_2d'], residual=sc)
        if tail:
            x = self._c(hip.scale_shift_relu(x.float(), net.bn1[0], net.bn1[1], relu=True))
        return x

    # ---- channels-last variant ---------------------------------------------------------
    # Activations are (H*W, C) rows (torch: NCHW-shaped tensors with channels_last strides), so a 1x1
    # convolution is rows x (Cin, Cout) with the bias + ReLU in the GEMM epilogue
    # (torch._addmm_activation -> hipBLASLt RELU_BIAS; in the NCHW form the bias runs along the GEMM's
    # other axis and needs its own pass), the library's NHWC MFMA convolutions are used without the
    # transposes it otherwise wraps around them, and the per-channel passes read float4 of channels.
    @staticmethod
    def _rows(x4):
        n, c, h, w = x4.shape
        # a view when x4 is channels_last-contiguous (what the library convolutions return for channels_last
        # inputs); callers only use the rows afterwards, so a copy here would still be correct
        return x4.permute(0, 2, 3, 1).reshape(n * h * w, c)

    @staticmethod
    def _map(x2, h, w):
        return x2.view(-1, h, w, x2.shape[1]).permute(0, 3, 1, 2)

    def _dcn_cl(self, c1_4, u, dilate):
        """DeformableConvolution on channels-last maps: offset conv, channels-last bilinear im2col (rows
        ordered (tap, channel)), one row GEMM with the weight permuted to match."""
        off = F.conv2d(c1_4, u['off_w_cl'], u['off_b'], stride=1, padding=dilate, dilation=dilate)
        col = hip.deform_im2col_cl(c1_4.permute(0, 2, 3, 1), off.permute(0, 2, 3, 1), 3, 3, dilate, 1, dilate,
                                   P.NUM_DEFORMABLE_GROUP)
        if self.taps is not None and _UNIT_TAPS:
            self.taps['u%d_%02d_1d_off' % (u['stage'], u['unit'])] = off
            self.taps['u%d_%02d_1e_col' % (u['stage'], u['unit'])] = col
        return torch.mm(col.view(-1, col.shape[2]), u['w2_tap_t'])     # (N*H*W, 9*C) x (9*C, Cout)

    def _resnet_cl(self, x, net, stages, tail, own_conv=False):
        """_resnet on channels-last activations (fp32).  Returns an NCHW-shaped channels_last map.
        own_conv: conv2 (3x3 + folded bn3 bias + ReLU) on lsfa_conv_nhwc_fwd instead of library conv + bias/ReLU pass."""
        net.prepare_channels_last()
        cl = torch.channels_last
        if 'stem' in _OWN_CONV and x.dtype == torch.float32 and net.conv0_w.dtype == torch.float32 and x.shape[1] == 3:
            # bn_data + conv0 + bn0 + relu0 and pool0 as two own launches (lsfa_stem_conv7x7s2, lsfa_maxpool3x3s2_nhwc)
            # instead of five library ones (affine pass, layout copy, convolution, bias + ReLU pass, max pool)
            if not hasattr(net, 'conv0_w_l'):
                net.conv0_w_l = hip.stem_weight_layout(net.conv0_w)
            y = hip.stem_conv(x, net.conv0_w_l, net.conv0_b, net.bn_data[0], net.bn_data[1])
            u0 = net.units[0]                                  # the first unit's bn1 + relu1 rides on the pooling launch
            pooled, a2_stem = hip.maxpool3x3s2_nhwc(y, scale2=u0['bn1'][0], shift2=u0['bn1'][1])
            x4 = pooled.permute(0, 3, 1, 2)
            stem_a2 = a2_stem.view(-1, a2_stem.shape[3])
        else:
            stem_a2 = None
            x = hip.scale_shift_relu(x, net.bn_data[0], net.bn_data[1], relu=False).contiguous(memory_format=cl)
            y = F.conv2d(x, net.conv0_w_cl, None, stride=2, padding=3)
            r = self._rows(y)
            hip.scale_shift_relu_cl(r, self._ones(r.shape[1]), net.conv0_b, relu=True, out=r)
            x4 = F.max_pool2d(self._map(r, y.shape[2], y.shape[3]), 3, 2, 1)
        dilate = 1
        units = [u for u in net.units if u['stage'] <= stages]
        fuse3 = own_conv and 'conv3' in _OWN_CONV
        a2 = stem_a2       # relu(bn1(x)) of the current unit when the producer of x already made it (stem / fused conv3)
        for ui, u in enumerate(units):
            first = u['unit'] == 1
            stride = 2 if (first and u['stage'] in (2, 3)) else 1
            unit_dilate = dilate
            if first and u['stage'] == 4:
                dilate = dilate * 2
            n, h, w = x4.shape[0], x4.shape[2], x4.shape[3]
            x2 = self._rows(x4)
            if a2 is None:
                a2 = hip.scale_shift_relu_cl(x2, u['bn1'][0], u['bn1'][1], relu=True)
            if own_conv and 'w1_split' in u:
                # the whole unit on the own split-bf16 convolution: conv1 (+ folded bn2 + ReLU), conv2 (3x3 or DCN: offsets,
                # bilinear columns, contraction; + folded bn3 + ReLU), the shortcut, conv3 + shortcut add in place + the
                # NEXT unit's bn1 + ReLU as a second output: no library GEMM, no separate BatchNorm pass
                taps = self.taps if (self.taps is not None and _UNIT_TAPS) else None
                tag = 'u%d_%02d_' % (u['stage'], u['unit'])
                a4 = a2.view(n, h, w, -1)
                c1 = hip.conv_split(a4, u['w1_split'], u['b1'], relu=True)
                if u['dcn']:
                    off = hip.conv_split(c1, u['off_split'], u['off_b_pad'], 1, unit_dilate, unit_dilate)
                    col = hip.deform_im2col_cl(c1, off, 3, 3, unit_dilate, 1, unit_dilate, P.NUM_DEFORMABLE_GROUP)
                    c2 = hip.conv_split(col.view(n, h, w, -1), u['dcn_split'], u['b2'], relu=True)
                    if taps is not None:
                        taps[tag + '1d_off'], taps[tag + '1e_col'] = off, col
                        if _os.environ.get('LSFA_DCN_CHECK') == '1':
                            # diagnostics: the same launch again right behind the first one, and what its inputs hold now
                            col2 = hip.deform_im2col_cl(c1, off, 3, 3, unit_dilate, 1, unit_dilate, P.NUM_DEFORMABLE_GROUP)
                            taps[tag + '1f_col_again'] = col2
                            taps[tag + '1g_col_neq_again'] = (col != col2).sum().float().view(1)
                            taps[tag + '1h_c1_then'] = c1.clone()
                            taps[tag + '1i_off_then'] = off.clone()
                else:
                    c2 = hip.conv_split(c1, u['w2_split'], u['b2'], stride, unit_dilate, unit_dilate, relu=True)
                ho, wo = c2.shape[1], c2.shape[2]
                sc4 = hip.conv_split(a4, u['sc_split'], None, stride) if first else x2.view(n, ho, wo, -1)
                nxt = units[ui + 1]['bn1'] if ui + 1 < len(units) else (net.bn1 if tail else None)
                if taps is not None:
                    taps[tag + '0a'], taps[tag + '1c1'], taps[tag + '2c2'] = a2, c1, c2
                    if first:
                        taps[tag + '3sc'] = sc4.clone()
                if nxt is not None:
                    _, out2 = hip.conv_split(c2, u['w3_split'], None, out=sc4, residual=sc4, out2=torch.empty_like(sc4),
                                             scale2=nxt[0], shift2=nxt[1])
                    a2 = out2.view(-1, out2.shape[3])
                else:
                    hip.conv_split(c2, u['w3_split'], None, out=sc4, residual=sc4)
                    a2 = None
                x4 = sc4.permute(0, 3, 1, 2)
                if taps is not None:
                    taps[tag + '4x'] = x4.clone()
                continue
            c1 = torch._addmm_activation(u['b1'], a2, u['w1_t'])                  # conv1 + folded bn2 + ReLU
            if self.taps is not None and _UNIT_TAPS:
                self.taps['u%d_%02d_0a' % (u['stage'], u['unit'])] = a2
                self.taps['u%d_%02d_1c1' % (u['stage'], u['unit'])] = c1
            dcn_own = u['dcn'] and own_conv and 'dcn_split' in u
            if dcn_own:
                # offsets (72 channels, padded to 128) by the own convolution, bilinear columns, contraction with bn3's bias + ReLU
                c14 = c1.view(n, h, w, -1)
                off = hip.conv_split(c14, u['off_split'], u['off_b_pad'], 1, unit_dilate, unit_dilate)
                col = hip.deform_im2col_cl(c14, off, 3, 3, unit_dilate, 1, unit_dilate, P.NUM_DEFORMABLE_GROUP)
                if 'dcn_split_h' in u:
                    # every column entry is a bilinear interpolation of c1 (zeros outside): max|col| <= max|c1|, 9x fewer bytes to scan
                    c2 = hip.conv_split_h(col.view(n, h, w, -1), u['dcn_split_h'], u['b2'], act=1, amax=hip.amax_partial(c14)).view(n * h * w, -1)
                else:
                    c2 = hip.conv_split(col.view(n, h, w, -1), u['dcn_split'], u['b2'], relu=True).view(n * h * w, -1)
                ho, wo = h, w
            elif u['dcn']:
                c2 = self._dcn_cl(self._map(c1, h, w), u, unit_dilate)
                ho, wo = h, w
            elif own_conv and 'w2_split' in u:
                y = hip.conv_split(c1.view(-1, h, w, c1.shape[1]), u['w2_split'], u['b2'], stride, unit_dilate, unit_dilate,
                                   relu=True)                      # folded bn3 bias + ReLU in the epilogue
                ho, wo = y.shape[1], y.shape[2]
                c2 = y.view(-1, y.shape[3])
            elif own_conv and 'w2_kc' in u:
                y = hip.conv_nhwc(c1.view(-1, h, w, c1.shape[1]), u['w2_kc'], u['b2'], 3, 3, stride, unit_dilate, unit_dilate,
                                  relu=True)                       # folded bn3 bias + ReLU in the epilogue
                ho, wo = y.shape[1], y.shape[2]
                c2 = y.view(-1, y.shape[3])
            else:
                c2_4 = F.conv2d(self._map(c1, h, w), u['w2_cl'], None, stride=stride, padding=unit_dilate,
                                dilation=unit_dilate)
                ho, wo = c2_4.shape[2], c2_4.shape[3]
                c2 = self._rows(c2_4)
            if (u['dcn'] and not dcn_own) or not (u['dcn'] or (own_conv and 'w2_kc' in u)):
                hip.scale_shift_relu_cl(c2, self._ones(c2.shape[1]), u['b2'], relu=True, out=c2)   # folded bn3 bias + ReLU
            if first:
                if stride != 1 and own_conv and 'sc_split' in u:
                    sc = hip.conv_split(a2.view(n, h, w, -1), u['sc_split'], None, stride).view(n * ho * wo, -1)
                elif stride != 1:
                    sc = self._rows(F.conv2d(self._map(a2, h, w), u['sc_cl'], None, stride=stride))
                else:
                    sc = torch.mm(a2, u['sc_t'])
            else:
                sc = x2                                         # overwritten in place by conv3 (+ the shortcut: beta = 1)
            if self.taps is not None and _UNIT_TAPS:
                self.taps['u%d_%02d_2c2' % (u['stage'], u['unit'])] = c2.clone()
                if first:
                    self.taps['u%d_%02d_3sc' % (u['stage'], u['unit'])] = sc.clone()
            # the bn1 + ReLU the NEXT unit (or the network's tail) applies to this unit's output
            nxt = units[ui + 1]['bn1'] if ui + 1 < len(units) else (net.bn1 if tail else None)
            if own_conv and 'conv3s' in _OWN_CONV and 'w3_split' in u and sc.is_contiguous():
                # conv3 + shortcut add in place + the next bn1 / ReLU as a second output on the split kernel; conv1 stays a library GEMM
                sc4 = sc.view(n, ho, wo, -1)
                if nxt is not None:
                    _, out2 = hip.conv_split(c2.view(n, ho, wo, -1), u['w3_split'], None, out=sc4, residual=sc4,
                                             out2=torch.empty_like(sc4), scale2=nxt[0], shift2=nxt[1])
                    a2 = out2.view(-1, out2.shape[3])
                else:
                    hip.conv_split(c2.view(n, ho, wo, -1), u['w3_split'], None, out=sc4, residual=sc4)
                    a2 = None
                x4 = self._map(sc, ho, wo)
            elif fuse3 and 'w3_kc' in u and sc.is_contiguous():
                # conv3 + shortcut add in place + the next bn1/relu1 as a second output: one launch (lsfa_conv_nhwc_fused_fwd)
                sc4 = sc.view(n, ho, wo, -1)
                out2 = torch.empty_like(sc4) if nxt is not None else None
                hip.conv_nhwc(c2.view(n, ho, wo, -1), u['w3_kc'], None, 1, 1, 1, 0, 1, relu=False, out=sc4, residual=sc4,
                              out2=out2, scale2=nxt[0] if nxt is not None else None, shift2=nxt[1] if nxt is not None else None)
                a2 = out2.view(-1, out2.shape[3]) if out2 is not None else None
                x4 = self._map(sc, ho, wo)
            else:
                x4 = self._map(sc.addmm_(c2, u['w3_t']), ho, wo)
                a2 = None
            if self.taps is not None and _UNIT_TAPS:
                self.taps['u%d_%02d_4x' % (u['stage'], u['unit'])] = x4.clone()
        if tail:
            h, w = x4.shape[2], x4.shape[3]
            if a2 is not None:
                x4 = self._map(a2, h, w)                          # the last conv3 already applied the final bn1 + relu1
            else:
                x4 = self._map(hip.scale_shift_relu_cl(self._rows(x4), net.bn1[0], net.bn1[1], relu=True), h, w)
        return x4

    def _backbone(self, data):
        if 'backbone' in self.cl:
            if not hasattr(self, 'feat_w_cl'):
                self.feat_w_cl = self.feat_w.contiguous(memory_format=torch.channels_last)
            x4 = self._resnet_cl(data, self.net, 4, True, own_conv='backbone' in _OWN_CONV)
            if _CONV_SPLIT and 'feat' in _OWN_CONV and self.feat_w.dtype == torch.float32:
                # feat_conv_3x3 (2048 -> 1024, dilation 6: the largest single convolution of a key frame) on the split-bf16
                # kernel: bias + ReLU in the epilogue, written in NCHW (what the warp / aggregation kernels and the API take)
                if 'h3' in _OWN_CONV:
                    # r3: two fp16 pieces per operand, three matrix instructions per product (lsfa_conv_split_h_fwd): as close to
                    # float64 as the bf16 three-piece form at half its matrix-pipe cycles; the scale comes from one pass over x4
                    if not hasattr(self, 'feat_w_split_h'):
                        self.feat_w_split_h = hip.SplitWeightH(self.feat_w)
                    return hip.conv_split_h(x4.permute(0, 2, 3, 1), self.feat_w_split_h, self.feat_b, 1, 6, 6, act=1, nchw=True)
                if not hasattr(self, 'feat_w_split'):
                    self.feat_w_split = hip.SplitWeight(self.feat_w)
                return hip.conv_split(x4.permute(0, 2, 3, 1), self.feat_w_split, self.feat_b, 1, 6, 6, relu=True, nchw=True)
            y = F.conv2d(x4, self.feat_w_cl, None, padding=6, dilation=6)
            r = self._rows(y)
            hip.scale_shift_relu_cl(r, self._ones(r.shape[1]), self.feat_b, relu=True, out=r)
            return self._map(r, y.shape[2], y.shape[3]).contiguous()   # NCHW for the warp / aggregation kernels and the API
        x = self._resnet(data, self.net, 4, True)
        return self._bias_act(F.conv2d(x, self.feat_w, None, padding=6, dilation=6).float(), self.feat_b.float(), True)

    def _flownet_prepare(self):
        """Weights of the own FlowNet path: split-bf16 fragments for the convolutions and the four phases of each
        Deconvolution(4x4, stride 2) + Crop(1) (input channels zero-padded to the concatenated maps' padded widths)."""
        fw, dev = self.flow, self.device
        own = {}
        w1 = fw['flow_conv1_weight']                                   # (64, 6, 7, 7)
        own['c1_cur'] = hip.stem_weight_layout(w1[:, 0:3].contiguous())
        own['c1_ref'] = hip.stem_weight_layout(w1[:, 3:6].contiguous())
        own['in_scale'] = torch.full((3,), 1.0 / 255.0, device=dev)
        own['in_shift'] = torch.zeros(3, device=dev)
        for name in ('conv2', 'conv3', 'conv3_1', 'conv4', 'conv4_1', 'conv5', 'conv5_1', 'conv6', 'conv6_1'):
            own[name] = hip.SplitWeight(fw[name + '_weight'])

        def pad32(c):
            return -(-c // 32) * 32
        for name in ('deconv5', 'deconv4', 'deconv3', 'deconv2'):
            wt = fw[name + '_weight']                                     # (Cin, Cout, 4, 4)
            own[name] = hip.deconv_phase_weights(wt, cin_pad=pad32(wt.shape[0]))
        ws = fw['Convolution5_scale_weight']                              # (1024, 194, 1, 1)
        wsp = torch.zeros((ws.shape[0], pad32(ws.shape[1]), 1, 1), device=dev, dtype=torch.float32)
        wsp[:, :ws.shape[1]] = ws
        own['scale'] = hip.SplitWeight(wsp, real_cin=ws.shape[1])
        for name in ('Convolution1', 'Convolution2', 'Convolution3', 'Convolution4', 'Convolution5'):
            own[name] = fw[name + '_weight'].permute(0, 2, 3, 1).contiguous()      # (2, 3, 3, Cin)
        self._flow_own = own

    def _flownet_own(self, img_cur, img_ref):
        """FlowNet-S (:150-207) on the own kernels, channels-last, one image pair: no library call.  Concat nodes are channel
        slices of maps the producers write into directly (lsfa_conv_split_view_fwd), Deconvolution + Crop are four phase
        convolutions, the 2-channel heads and their upsampling are small dedicated kernels (csrc/flownet.hip)."""
        if not hasattr(self, '_flow_own'):
            self._flownet_prepare()
        o, fw, dev = self._flow_own, self.flow, self.device
        LEAKY = 2

        def cmap(h, w, c):      # a concatenated map with its channel count padded to a multiple of 32 (the padding stays zero)
            return torch.zeros((1, h, w, -(-c // 32) * 32), device=dev, dtype=torch.float32)

        def out_hw(h, w, k, stride, pad):
            return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1

        def conv(x, name, stride, pad, out=None, c0=0, cin=None):
            sw = o[name]
            h, w = out_hw(x.shape[1], x.shape[2], sw.kh, stride, pad)
            if out is None:
                out = torch.empty((1, h, w, sw.cout), device=dev, dtype=torch.float32)
            return hip.conv_split_view(x, sw, fw[name + '_bias'], out, stride=stride, pad=(pad, pad), act=LEAKY, cin=cin, c0=c0)

        def deconv(x, name, out, c0):
            hip.deconv4x4s2_crop(x, o[name], fw[name + '_bias'], out, c0=c0, act=LEAKY)

        def head(x, name, cin):
            return hip.head_conv3x3(x, o[name], fw[name + '_bias'], cin=cin)

        # avg pool 2x2 of each image (x 1/255 folded into the first convolution's input affine), flow_conv1 as two passes
        pc, pr = hip.avgpool_nchw(img_cur, 2), hip.avgpool_nchw(img_ref, 2)
        r1 = hip.stem_conv(pc, o['c1_cur'], None, o['in_scale'], o['in_shift'], act=0)
        r1 = hip.stem_conv(pr, o['c1_ref'], fw['flow_conv1_bias'], o['in_scale'], o['in_shift'], out=r1, accum=r1, act=LEAKY)
        h2, w2 = out_hw(r1.shape[1], r1.shape[2], 5, 2, 2)
        c5 = cmap(h2, w2, 194)
        conv(r1, 'conv2', 2, 2, out=c5)                                   # r2 = c5[..., :128]
        r3 = conv(c5, 'conv3', 2, 2, cin=128)
        c4 = cmap(r3.shape[1], r3.shape[2], 386)
        conv(r3, 'conv3_1', 1, 1, out=c4)                                 # r4 = c4[..., :256]
        r5 = conv(c4, 'conv4', 2, 1, cin=256)
        c3 = cmap(r5.shape[1], r5.shape[2], 770)
        conv(r5, 'conv4_1', 1, 1, out=c3)                                 # r6 = c3[..., :512]
        r7 = conv(c3, 'conv5', 2, 1, cin=512)
        c2 = cmap(r7.shape[1], r7.shape[2], 1026)
        conv(r7, 'conv5_1', 1, 1, out=c2)                                 # r8 = c2[..., :512]
        r9 = conv(c2, 'conv6', 2, 1, cin=512)
        r10 = conv(r9, 'conv6_1', 1, 1)
        f6 = head(r10, 'Convolution1', 1024)
        deconv(r10, 'deconv5', c2, 512)
        hip.upsample_flow(f6, fw['upsample_flow6to5_weight'], fw['upsample_flow6to5_bias'], c2, 1024)
        f5 = head(c2, 'Convolution2', 1026)
        deconv(c2, 'deconv4', c3, 512)
        hip.upsample_flow(f5, fw['upsample_flow5to4_weight'], fw['upsample_flow5to4_bias'], c3, 768)
        f4 = head(c3, 'Convolution3', 770)
        deconv(c3, 'deconv3', c4, 256)
        hip.upsample_flow(f4, fw['upsample_flow4to3_weight'], fw['upsample_flow4to3_bias'], c4, 384)
        f3 = head(c4, 'Convolution4', 386)
        deconv(c4, 'deconv2', c5, 128)
        hip.upsample_flow(f3, fw['upsample_flow3to2_weight'], fw['upsample_flow3to2_bias'], c5, 192)
        c5p = hip.avgpool2_nhwc(c5)
        flow = hip.head_conv3x3(c5p, o['Convolution5'], fw['Convolution5_bias'], cin=194, mul=2.5, nchw=True)
        scale = hip.conv_split(c5p, o['scale'], fw['Convolution5_scale_bias'], nchw=True)
        return flow, scale

    def _flownet(self, img_cur, img_ref):
        """FlowNet-S on the half-resolution pair (:150-207)."""
        if _CONV_SPLIT and 'flow' in _OWN_CONV and self.cdtype == torch.float32 and img_cur.shape[0] == 1 and img_cur.is_cuda:
            return self._flownet_own(img_cur, img_ref)
        fw = self.flow

        fused = self.cdtype == torch.float32      # bias + LeakyReLU as one HIP pass (fp32 maps)
        gemm_max = _FLOW_GEMM_MAX_L if fused else 0

        def bias_act(y, b, act):
            if fused:
                c = y.shape[1]
                if act:
                    return hip.scale_shift_leaky(y, self._ones(c), b, 0.1, out=y)
                return hip.scale_shift_relu(y, self._ones(c), b, relu=False, out=y)
            y = y + b.view(1, -1, 1, 1)
            return F.leaky_relu_(y, 0.1) if act else y

        def conv(x, name, stride=1, pad=1, act=True):
            w, b = fw[name + '_weight'], fw[name + '_bias']
            k = w.shape[2]
            ho, wo = (x.shape[2] + 2 * pad - k) // stride + 1, (x.shape[3] + 2 * pad - k) // stride + 1
            if x.shape[0] == 1 and ho * wo <= gemm_max:
                # small output map: the weights dominate the traffic (conv6_1: 37.7 MB of weights for 40
                # output pixels); im2col + one weight-streaming GEMM beats the library's convolution kernels
                col = F.unfold(x, k, padding=pad, stride=stride)[0]               # (Cin*k*k, ho*wo)
                y = torch.mm(w.view(w.shape[0], -1), col).view(1, -1, ho, wo)
            else:
                y = F.conv2d(x, w, None, stride=stride, padding=pad)
            return bias_act(y, b, act)

        def deconv(x, name, like, act):
            y = F.conv_transpose2d(x, fw[name + '_weight'], None, stride=2)
            y = bias_act(y, fw[name + '_bias'], act)                              # on the full map, then
            return y[:, :, 1:1 + like.shape[2], 1:1 + like.shape[3]]              # Crop(offset=(1,1)) to the skip tensor

        data = self._c(torch.cat([img_cur / 255.0, img_ref / 255.0], 1))
        x = F.avg_pool2d(data, 2, 2, ceil_mode=True)
        r1 = conv(x, 'flow_conv1', 2, 3)
        r2 = conv(r1, 'conv2', 2, 2)
        r3 = conv(r2, 'conv3', 2, 2)
        r4 = conv(r3, 'conv3_1')
        r5 = conv(r4, 'conv4', 2)
        r6 = conv(r5, 'conv4_1')
        r7 = conv(r6, 'conv5', 2)
        r8 = conv(r7, 'conv5_1')
        r9 = conv(r8, 'conv6', 2)
        r10 = conv(r9, 'conv6_1')
        f6 = conv(r10, 'Convolution1', act=False)
        c2 = torch.cat([r8, deconv(r10, 'deconv5', r8, True), deconv(f6, 'upsample_flow6to5', r8, False)], 1)
        f5 = conv(c2, 'Convolution2', act=False)
        c3 = torch.cat([r6, deconv(c2, 'deconv4', r6, True), deconv(f5, 'upsample_flow5to4', r6, False)], 1)
        f4 = conv(c3, 'Convolution3', act=False)
        c4 = torch.cat([r4, deconv(c3, 'deconv3', r4, True), deconv(f4, 'upsample_flow4to3', r4, False)], 1)
        f3 = conv(c4, 'Convolution4', act=False)
        c5 = torch.cat([r2, deconv(c4, 'deconv2', r2, True), deconv(f3, 'upsample_flow3to2', r2, False)], 1)
        c5 = F.avg_pool2d(c5, 2, 2, ceil_mode=True)
        flow = conv(c5, 'Convolution5', act=False).float() * 2.5
        scale = conv(c5, 'Convolution5_scale', pad=0, act=False).float()
        return flow, scale

    def _heads(self, conv_feat, im_info):
        """SliceChannel -> RPN -> Proposal -> R-FCN maps -> PSROI + average + softmax (:479-546)."""
        cfg = self.cfg
        A = cfg.network.NUM_ANCHORS
        n, _, h, w = conv_feat.shape
        rpn = self._conv1x1(self._c(conv_feat[:, :512]), self.rpn_w, bias=self.rpn_b).float()
        cls_prob = torch.softmax(rpn[:, :2 * A].reshape(n, 2, A * h, w), dim=1).reshape(n, 2 * A, h, w)
        rois = self.proposal(cls_prob, rpn[:, 2 * A:], im_info)
        if self.ps_layout and n == 1 and _CONV_SPLIT and 'rfcn' in _OWN_CONV and self.cdtype == torch.float32 and conv_feat.is_cuda:
            # r3 (opt-in, LSFA_OWN_CONV=...,rfcn): the same GEMM on the own split convolution ((H*W, 512) x (512, 49*D), D = ncls + nbox,
            # columns padded to a multiple of 128): the R-FCN half of conv_feat is turned channels-last by one small launch, the
            # padded position-sensitive map is read with its row stride by the head kernel.  Measured 44.6 us (128 x 128 tiles) + 9.2
            # (transpose) against 45 us for the tuned hipBLASLt fp32-MFMA GEMM (104 TFLOP/s, 66 % of ITS pipe's peak): no gain, so the
            # library GEMM stays the default here
            D = self.ncls + self.nbox
            if not hasattr(self, '_rfcn_split'):
                cols = 49 * D
                pad_to = -(-cols // 128) * 128
                wp = torch.zeros((pad_to, 512, 1, 1), device=self.device, dtype=torch.float32)
                wp[:cols, :, 0, 0] = self.rfcn_w_ps_t.t()
                bp = torch.zeros(pad_to, device=self.device, dtype=torch.float32)
                bp[:cols] = self.rfcn_b_ps
                self._rfcn_split, self._rfcn_b_pad, self._rfcn_ld = hip.SplitWeight(wp, real_cout=cols), bp, pad_to
            x = hip.nchw_to_nhwc(conv_feat, 512, 512)
            ps = hip.conv_split(x, self._rfcn_split, self._rfcn_b_pad)                 # (1, h, w, ld)
            if self.taps is not None:
                nchw = ps.view(h * w, self._rfcn_ld)[:, :49 * D].reshape(h * w, 49, D).permute(2, 1, 0).reshape(1, D * 49, h, w)
                self.taps.update(rpn_cls_prob=cls_prob, rpn_bbox_pred=rpn[:, 2 * A:], cls_map=nchw[:, :self.n_cls_ch],
                                 box_map=nchw[:, self.n_cls_ch:])
            cls_p, bbox = hip.rfcn_head_ps_ld(ps, self._rfcn_ld, rois, h, w, self.ncls, self.nbox, 0.0625, 7, 7)
        elif self.ps_layout and n == 1:
            # both R-FCN convs as one GEMM that writes the position-sensitive layout directly
            xt = self._c(conv_feat[0, 512:]).view(512, h * w).t()
            D = self.ncls + self.nbox
            ps = torch.addmm(self.rfcn_b_ps, xt, self.rfcn_w_ps_t).float().view(1, h, w, 49, D)
            if self.taps is not None:
                nchw = ps.view(h * w, 49, D).permute(2, 1, 0).reshape(1, D * 49, h, w)
                self.taps.update(rpn_cls_prob=cls_prob, rpn_bbox_pred=rpn[:, 2 * A:], cls_map=nchw[:, :self.n_cls_ch],
                                 box_map=nchw[:, self.n_cls_ch:])
            cls_p, bbox = hip.rfcn_head_ps(ps, rois, self.ncls, self.nbox, 0.0625, 7, 7)
        else:
            maps = self._conv1x1(self._c(conv_feat[:, 512:]), self.rfcn_w, bias=self.rfcn_b).float()
            if self.taps is not None:
                self.taps.update(rpn_cls_prob=cls_prob, rpn_bbox_pred=rpn[:, 2 * A:], cls_map=maps[:, :self.n_cls_ch],
                                 box_map=maps[:, self.n_cls_ch:])
            cls_p, bbox = hip.rfcn_head(maps[:, :self.n_cls_ch], maps[:, self.n_cls_ch:], rois, 0.0625, 7, 7)
        B = cfg.TEST.BATCH_IMAGES
        return rois, cls_p.view(B, -1, cls_p.shape[1]), bbox.view(B, -1, bbox.shape[1])

    # ---- forward ---------------------------------------------------------------------
    def forward(self, **inputs):
        with torch.no_grad():
            if self.sym.kind == 'batch':
                return self._forward_batch(inputs)
            return self._forward_key(inputs) if self.sym.kind == 'key' else self._forward_cur(inputs)

    def _forward_batch(self, d):
        """get_batch_test_symbol (:661-751): one key frame + N other frames in one pass.  tile_as
        (operator_py/tile_as.py:16-19) is a broadcast view for FlowNet's reference image and the
        warp kernel's feat_n = 1 mode for the key feature (no (N,1024,h,w) copy)."""
        data_key, data_other = d['data_key'], d['data_other']
        n = data_other.shape[0]
        conv_feat_key = self._backbone(data_key)
        flow, scale_map = self._flownet(data_other, data_key.expand(n, -1, -1, -1))
        conv_feat_other = hip.warp_bilinear(conv_feat_key, flow, mul=scale_map)
        if self.taps is not None:
            self.taps.update(backbone_feat=conv_feat_key, flow=flow, scale_map=scale_map, warp=conv_feat_other)
        conv_feat = torch.cat([conv_feat_key, conv_feat_other], 0)
        rois, cls_prob, bbox_pred = self._heads(conv_feat, d['im_info'])
        return {'rois_output': rois, 'cls_prob_reshape_output': cls_prob, 'bbox_pred_reshape_output': bbox_pred}

    def _forward_key(self, d):
        cfg = self.cfg
        data, feat_key_old = d['data'], d['feat_key_old']
        # ChooseOldKeyFeat: first frame <=> placeholder shape (1, 1024, 1, 1)
        _, c, h, w = feat_key_old.shape
        is_first = (c == cfg.network.DFF_FEAT_DIM and h == 1 and w == 1)
        conv_feat, flow, scale_map = self._key_front(data, None if is_first else d['data_key_old'])
        out = self._key_back(conv_feat, flow, scale_map, feat_key_old, d['im_info'])
        out.update({'data_key': d.get('data_key'), 'motion_vector': d.get('motion_vector'), 'res_diff': d.get('res_diff'),
                    'feat_key': d.get('feat_key')})
        return out

    def key_front(self, data, data_key_old):
        """The part of a key frame that does not depend on the previous key frame's FEATURE: backbone of
        this frame and FlowNet(this frame, previous key image).  -> (conv_feat, flow, scale_map).
        lsfa_amd/core/graphs.py runs it ahead of time, beside the previous key frame."""
        with torch.no_grad():
            return self._key_front(data, data_key_old)

    def key_backbone(self, data):
        """key_front's two independent halves, for callers that run them on different streams."""
        with torch.no_grad():
            conv_feat = self._backbone(data)
            self._tap('backbone_feat', conv_feat)
            return conv_feat

    def key_flow(self, data, data_key_old):
        with torch.no_grad():
            return self._flownet(data, data_key_old)

    def key_back(self, conv_feat, flow, scale_map, feat_key_old, im_info):
        """The rest of the key frame: flow warp x scale map of the old key feature, aggregation, heads."""
        with torch.no_grad():
            return self._key_back(conv_feat, flow, scale_map, feat_key_old, im_info)

    def _key_front(self, data, data_key_old):
        conv_feat = self._backbone(data)
        self._tap('backbone_feat', conv_feat)
        if data_key_old is None:
            return conv_feat, None, None
        flow, scale_map = self._flownet(data, data_key_old)
        return conv_feat, flow, scale_map

    def _key_back(self, conv_feat, flow, scale_map, feat_key_old, im_info):
        conv_feat = self._key_aggregate(conv_feat, flow, scale_map, feat_key_old)
        return self._key_heads(conv_feat, im_info)

    def key_aggregate(self, conv_feat, flow, scale_map, feat_key_old):
        """key_back's two halves, for callers that run the heads elsewhere: the aggregated feature ..."""
        with torch.no_grad():
            return self._key_aggregate(conv_feat, flow, scale_map, feat_key_old)

    def key_heads(self, conv_feat, im_info):
        """... and RPN + Proposal + R-FCN heads on it."""
        with torch.no_grad():
            return self._key_heads(conv_feat, im_info)

    def _key_aggregate(self, conv_feat, flow, scale_map, feat_key_old):
        cfg = self.cfg
        if flow is not None:
            warp = hip.warp_bilinear(feat_key_old, flow, mul=scale_map)
            if self.taps is not None:
                self.taps.update(flow=flow, scale_map=scale_map, warp=warp)
            if cfg.network.add_Nq_net and _CONV_SPLIT and 'nq' in _OWN_CONV and self.cdtype == torch.float32 and warp.shape[0] == 1:
                logits = self._nq_own(warp, conv_feat)
                self._tap('nq_logits', logits)
                conv_feat = hip.aggregate_softmax2(warp, conv_feat, logits)
            elif cfg.network.add_Nq_net:
                x = self._c(torch.cat([warp, conv_feat], 0))
                x = torch.relu_(F.conv2d(x, self.nq[0][0], self.nq[0][1], padding=1))
                x = torch.relu_(F.conv2d(x, self.nq[1][0], self.nq[1][1]))
                logits = F.conv2d(x, self.nq[2][0], self.nq[2][1]).float()
                self._tap('nq_logits', logits)
                conv_feat = hip.aggregate_softmax2(warp, conv_feat, logits)
            elif cfg.network.add_Fgfa_net:
                x = self._c(torch.cat([conv_feat, warp], 0))          # note the order, :133
                x = torch.relu_(F.conv2d(x, self.em[0][0], self.em[0][1]))
                x = torch.relu_(F.conv2d(x, self.em[1][0], self.em[1][1], padding=1))
                e = F.conv2d(x, self.em[2][0], self.em[2][1]).float()
                self._tap('embed', e)
                conv_feat = hip.aggregate_cosine(warp, conv_feat, e[1:2], e[0:1])
            else:
                conv_feat = 0.5 * (warp + conv_feat)
        return conv_feat

    def _nq_own(self, warp, conv_feat):
        """Nq_net (:94-109) on the own convolution: the two maps side by side as a batch of 2 channels-last images, 3x3
        1024 -> 256 + ReLU, 1x1 256 -> 16 + ReLU, 1x1 16 -> 1; the 16 / 1 output channels are padded to the kernel's 64-channel
        tiles with zero weights (the padded activations are relu(0) = 0 and meet zero weights again).  -> logits (2, 1, H, W)."""
        if not hasattr(self, '_nq_split'):
            dev = self.device
            (w1, b1), (w2, b2), (w3, b3) = self.nq

            def padded(w, b, cin_to, cout_to):
                wp = torch.zeros((cout_to, cin_to) + tuple(w.shape[2:]), device=dev, dtype=torch.float32)
                wp[:w.shape[0], :w.shape[1]] = w
                bp = torch.zeros(cout_to, device=dev, dtype=torch.float32)
                bp[:b.shape[0]] = b
                return hip.SplitWeight(wp, real_cout=w.shape[0], real_cin=w.shape[1]), bp
            self._nq_split = [(hip.SplitWeight(w1), b1), padded(w2, b2, w2.shape[1], 64), padded(w3, b3, 64, 64)]
        (s1, b1), (s2, b2), (s3, b3) = self._nq_split
        _, c, h, w = warp.shape
        x = torch.empty((2, h, w, c), device=warp.device, dtype=torch.float32)
        x[0].copy_(warp[0].permute(1, 2, 0))            # NCHW -> channels-last, both maps into one batch
        x[1].copy_(conv_feat[0].permute(1, 2, 0))
        x = hip.conv_split(x, s1, b1, 1, 1, 1, relu=True)
        x = hip.conv_split(x, s2, b2, relu=True)
        x = hip.conv_split(x, s3, b3)
        return x[..., 0].reshape(2, 1, h, w).contiguous()

    def _key_heads(self, conv_feat, im_info):
        rois, cls_prob, bbox_pred = self._heads(conv_feat, im_info)
        return {'choose_feat_output': conv_feat, 'rois_output': rois, 'cls_prob_reshape_output': cls_prob,
                'bbox_pred_reshape_output': bbox_pred}

    def small_net_feature(self, data):
        """fuse_small_net's image branch (:209-236): avgpool 4x4 -> small_net_ stem + stage 1 ->
        fuse_reduce_add.  It depends on the frame image only, so a caller may compute it ahead of the
        rest of the frame (lsfa_amd/core/graphs.py overlaps it with the previous frame's tail)."""
        with torch.no_grad():
            if 'stem' in _OWN_CONV and 'small' in self.cl and data.dtype == torch.float32:
                img = hip.avgpool_nchw(data, 4)
            else:
                img = F.avg_pool2d(data, 4, 4, ceil_mode=True)
            if 'small' in self.cl:
                if not hasattr(self, 'fuse_w_cl'):
                    self.fuse_w_cl = self.fuse_w.contiguous(memory_format=torch.channels_last)
                own = 'small' in _OWN_CONV
                s = self._resnet_cl(img, self.small, 1, False, own_conv=own)
                if _CONV_SPLIT and self.fuse_w.dtype == torch.float32 and 'fuse' in _OWN_CONV:
                    if 'h3' in _OWN_CONV:
                        if not hasattr(self, 'fuse_w_split_h'):
                            self.fuse_w_split_h = hip.SplitWeightH(self.fuse_w)
                        return hip.conv_split_h(s.permute(0, 2, 3, 1), self.fuse_w_split_h, self.fuse_b, 1, 1, 1, act=0, nchw=True)
                    if not hasattr(self, 'fuse_w_split'):
                        self.fuse_w_split = hip.SplitWeight(self.fuse_w)
                    # written in NCHW by the epilogue: the warp kernel's `add` operand, no transposing copy
                    return hip.conv_split(s.permute(0, 2, 3, 1), self.fuse_w_split, self.fuse_b, 1, 1, 1, relu=False, nchw=True)
                if own:
                    if not hasattr(self, 'fuse_w_kc'):
                        self.fuse_w_kc = hip.conv_weight_kc(self.fuse_w)
                    y = hip.conv_nhwc(s.permute(0, 2, 3, 1), self.fuse_w_kc, self.fuse_b, 3, 3, 1, 1, 1, relu=False)
                    return y.permute(0, 3, 1, 2).contiguous()        # NCHW for the warp kernel's `add` operand
                return F.conv2d(s, self.fuse_w_cl, self.fuse_b, padding=1).contiguous()
            s = self._resnet(img, self.small, 1, False)
            return F.conv2d(s, self.fuse_w, self.fuse_b, padding=1).float()

    def _forward_cur(self, d):
        cfg = self.cfg
        add = d.get('small_feat')          # precomputed by the caller, else computed here
        if add is None and cfg.network.add_small_net:
            add = self.small_net_feature(d['data'])
        self._tap('small_feat', add)
        conv_feat = hip.warp_bilinear(d['feat_key'], d['motion_vector'], add=add, res=d['res_diff'], res_w=self.rnet_w,
                                      res_b=self.rnet_b)
        rois, cls_prob, bbox_pred = self._heads(conv_feat, d['im_info'])
        return {'data': d['data'], 'data_key': d.get('data_key'), 'data_key_old': d.get('data_key_old'),
                'feat_key_old': d.get('feat_key_old'), 'rois_output': rois, 'cls_prob_reshape_output': cls_prob,
                'bbox_pred_reshape_output': bbox_pred, 'conv_feat': conv_feat}
