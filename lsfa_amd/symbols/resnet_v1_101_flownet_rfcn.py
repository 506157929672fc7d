"""LSFA test graphs (key-frame and non-key-frame) on MI355X.

API mirror of dff_rfcn/symbols/resnet_v1_101_flownet_rfcn.py: the class
`resnet_v1_101_flownet_rfcn(cfg)` with `get_key_test_symbol(cfg)` (:448-551),
`get_cur_test_symbol(cfg)` (:553-659), `get_batch_test_symbol(cfg)` (:661-751) and
`init_weight(cfg, arg_params, aux_params)` (:753-870).
A "symbol" here is a `TestSymbol`: it knows its argument / auxiliary / output names and
shapes like an mx.sym.Group, and `bind()`s to an executor that runs the graph.

Execution design (not a translation of the MXNet graph) - ONE path per dtype since r4:
  * every convolution of ResNet-101 (+ DCN), the small net, FlowNet-S, the Nq / embedding nets and feat_conv_3x3 runs on the
    hand-written split-operand MFMA kernels behind lsfa_conv_fwd (lsfa_amd/csrc/conv_ring_kernel.h, conv_split_kernel.h),
    channels-last, fp32 in / fp32 accumulate / fp32 out.  `dtype=torch.float32` (BASELINE configs[1]) forms every fp32 product from
    two fp16 pieces per operand and a power-of-two scale per map (three matrix instructions per product; FlowNet too: a Concat map's
    producers share one row of amax slots); `dtype=torch.bfloat16` (configs[2]) rounds the operands to one bf16 piece (one product).
    No library convolution or GEMM in either mode;
  * inference BatchNorms are folded at bind time: bn2 / bn3 of each pre-activation unit and bn0 into the preceding convolution;
    each unit's bn1 + relu1 (its input is also the raw shortcut, so it cannot be folded) is the SECOND OUTPUT of the previous
    unit's conv3 epilogue (y = conv3 + shortcut in place, y2 = relu(bn1_next(y))), the first one rides on the max-pooling launch;
    the residual add is conv3's epilogue; rpn_inv_normalize (operator_py/rpn_inv_normalize.py:19-26) is folded into
    rpn_bbox_pred's weights;
  * the scale of the fp16 form is max|x| of the input map, left behind by the PRODUCING convolution's epilogue (`amax_out`, 256
    atomicMax slots zeroed once per frame section) - no pass of its own; an under-estimated scale raises the executor's status
    word (`Executor.check_status()`), it cannot silently overflow;
  * the heads read the NCHW feature the reference's operators exchange: both RPN convs are one convolution of the own family on the NCHW
    map itself (x_nchw: the direct kernel's K-major operand form) + lsfa_rpn_softmax_split; both R-FCN convs are ONE convolution of the own family on a channels-last copy of channels 512.. (lsfa_nchw_to_nhwc,
    which also leaves their maximum) that writes the position-sensitive layout lsfa_rfcn_head_ps_ld_fwd reads in place;
  * every pass takes a batch: B lock-step clips, the F non-key frames of a segment (x B, frame-major: image f * B + b is warped from clip
    b's key feature) or the fronts of G key frames - what lsfa_amd/core/graphs.py FramePipeline(segment, key_group) feeds it;
  * warp, x scale_map, + rnet_conv0(res_diff), + small-net feature, the Nq softmax-combine, Proposal, PSROI pooling + 7x7 average +
    class softmax, and DCN's bilinear im2col are the hand-written HIP kernels behind include/lsfa_hip.h;
  * ChooseOldKeyFeat / ChooseFeat (operator_py/choose_old_key_feat.py:23-32, choose_feat.py:23-31) are a host-side `if` on the
    SHAPE of feat_key_old, exactly the reference's first-frame test, without its GPU->CPU sync; on the first frame FlowNet / warp /
    Nq are skipped because ChooseFeat discards their result.
"""
import os as _os

import numpy as np
import torch

from lsfa_amd import hip
from lsfa_amd.symbols import params as P

BN_EPS = 2e-5  # sym_common.py:9
# lab: pieces per fp32 operand of the fp32 path's convolutions (2 = fp16 hi / lo + scale, the default; 3 = three bf16 pieces)
_FP32_PIECES = int(_os.environ.get('LSFA_CONV_PIECES', '2'))


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

    def bind(self, arg_params, aux_params, device='cuda:0', dtype=torch.float32, pieces=None):
        """pieces (r5; fp32 only): None = the default form (two fp16 pieces per product, or LSFA_CONV_PIECES); 0 = the EXACT fp32 evaluation,
        every product an fp32 matrix-instruction product (lsfa_conv_nhwc_fused_fwd) - a slow reference for A/B tests and bench.py's
        `value_exact_fp32`, see Executor."""
        return Executor(self, arg_params, aux_params, device, dtype, pieces)


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


def _pad_rows(w, b, cout_to, cin_to=None):
    """zero-pad a (Cout, Cin, kh, kw) weight (and its bias) to the kernels' channel granularities: the padded outputs are
    act(0 + 0) = 0 and meet zero weights in the next layer"""
    cin_to = w.shape[1] if cin_to is None else cin_to
    wp = torch.zeros((cout_to, cin_to) + tuple(w.shape[2:]), device=w.device, dtype=torch.float32)
    wp[:w.shape[0], :w.shape[1]] = w
    bp = None
    if b is not None:
        bp = torch.zeros(cout_to, device=w.device, dtype=torch.float32)
        bp[:b.shape[0]] = b
    return wp, bp


class _ResNetWeights(object):
    """Folded, cut and laid-out weights of a (prefix) pre-activation ResNet-101 (or its stem + stage 1): resnet.py:138-240."""

    def __init__(self, arg, aux, prefix, stages, add_dcn, tail, device, pieces):
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

        def dev(a):
            return _t(np.asarray(a, np.float32), device, f32)

        def split(w, **kw):
            return hip.SplitWeight(dev(w), pieces=pieces, **kw)

        s, t = bn(prefix + 'bn_data')
        self.bn_data = (dev(s), dev(t))
        s0, t0 = bn(prefix + 'bn0')
        self.conv0_w_l = hip.stem_weight_layout(dev(fold(arg[prefix + 'conv0_weight'], s0)))
        self.conv0_b = dev(t0)
        self.conv0_exact = None
        if pieces == 0:      # exact-fp32 reference mode: conv0 as a 7x7 / stride 2 convolution on the input padded to 32 channels
            w0 = np.zeros((64, 32, 7, 7), np.float32)
            w0[:, :3] = fold(arg[prefix + 'conv0_weight'], s0)
            self.conv0_exact = hip.SplitWeight(dev(w0), real_cin=3, pieces=0)
        self.units = []
        for si in range(1, stages + 1):
            for u in range(1, P.UNITS[si - 1] + 1):
                p = '%sstage%d_unit%d_' % (prefix, si, u)
                s1, t1 = bn(p + 'bn1')
                s2, t2 = bn(p + 'bn2')
                s3, t3 = bn(p + 'bn3')
                w2 = fold(arg[p + 'conv2_weight'], s3)
                d = dict(stage=si, unit=u, bn1=(dev(s1), dev(t1)), dcn=P.is_dcn_unit(si, u, add_dcn),
                         w1=split(fold(arg[p + 'conv1_weight'], s2)), b1=dev(t2), b2=dev(t3), w3=split(arg[p + 'conv3_weight']))
                if u == 1:
                    d['sc'] = split(arg[p + 'sc_weight'])
                if d['dcn']:
                    # offset branch (sym_common.py:249-257): 72 channels, zero-padded to the kernels' 64-channel tiles; the contraction
                    # of the sampled columns is a 1x1 convolution over 9 * C "channels" ordered (tap, c) like lsfa_deform_im2col_cl's rows
                    ow, ob = dev(arg[p + 'conv2_offset_weight']), dev(arg[p + 'conv2_offset_bias'])
                    co = ow.shape[0]
                    wp, bp = _pad_rows(ow, ob, -(-co // 64) * 64)
                    d['off'], d['off_b'] = hip.SplitWeight(wp, real_cout=co, pieces=pieces), bp
                    w2t = dev(w2)
                    d['w2'] = hip.SplitWeight(w2t.permute(0, 2, 3, 1).reshape(w2t.shape[0], -1, 1, 1).contiguous(), pieces=pieces)
                else:
                    d['w2'] = split(w2)
                self.units.append(d)
        self.bn1 = None
        if tail:
            s, t = bn(prefix + 'bn1')
            self.bn1 = (dev(s), dev(t))


CUR_CHANNELS_LAST = _os.environ.get('LSFA_CUR_NCHW') != '1'


class _Slots(object):
    """amax_out slot rows of one section of a frame: a fresh zeroed (rows, 256) block per call (inside a hipGraph capture that
    is graph-private memory, so graphs replayed side by side - the non-key lanes - never share slots)."""

    def __init__(self, rows, device, enabled):
        self.rows, self.device, self.enabled, self.t, self.n = rows, device, enabled, None, 0

    def begin(self):
        if self.enabled:
            self.t, self.n = hip.amax_slots(self.rows, self.device), 0
        return self

    def new(self):
        if not self.enabled:
            return None
        if self.n >= self.rows:
            raise hip.LsfaError("amax slots: section needs more than %d rows" % self.rows)
        self.n += 1
        return self.t[self.n - 1]


class Executor(object):
    """A bound test symbol: folded weights on the device + forward()."""

    def __init__(self, sym, arg_params, aux_params, device, dtype, pieces=None):
        self.sym, self.cfg = sym, sym.cfg
        self.device = torch.device(device)
        self.cdtype = dtype           # fp32: split-operand convolutions with fp32 accuracy; bf16: one bf16 product per fp32 product
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("Executor: dtype must be torch.float32 or torch.bfloat16")
        # pieces per fp32 operand.  0 (r5) = exact fp32 products: every convolution of the ResNets, the DCN branch, feat_conv_3x3, the small net,
        # fuse_reduce_add, the Nq / embedding nets, the RPN and R-FCN heads and FlowNet's plain convolutions run on v_mfma_f32_32x32x2_f32
        # (hip._conv_exact); FlowNet's first layer (the stem kernel's two fp16 pieces) and its four transposed convolutions (three exact
        # bf16 pieces) keep their kernels.  A reference evaluation, ~6x slower.
        self.pieces = (_FP32_PIECES if pieces is None else int(pieces)) if dtype == torch.float32 else 1
        self.flow_pieces = self.pieces       # FlowNet too: a Concat map's scale is the maximum over its producers (they share its amax slots)
        self.taps = None              # set to {} to record stage outputs (parity tests)
        self.status = hip.new_status(self.device)
        # conv1 applies its unit's bn1 + relu1 where it cuts its operand (see _resnet); 0: every conv3 stores the activated map (r3's form, for A/B runs)
        self.input_activation_at_cut = _os.environ.get('LSFA_INPUT_ACT_AT_CUT', '1') == '1'
        # ... where the map is large: below ~48 MB (a single 1000x600 image past stage 1) the maps live in L2 / the Infinity Cache, conv3's
        # second output costs little and conv1's extra arithmetic shows (tools/lab: 3088 vs 3160 us for one image, 10706 vs 11080 for six).
        # The choice never changes a bit of the result (tests/test_hip_ops.py::test_conv_input_activation_at_the_cut).
        self.input_activation_min_bytes = int(float(_os.environ.get('LSFA_INPUT_ACT_MIN_MB', '48')) * (1 << 20))
        cfg = self.cfg
        arg = {k: np.asarray(v, dtype=np.float32) for k, v in arg_params.items()}
        aux = {k: np.asarray(v, dtype=np.float32) for k, v in aux_params.items()}
        for k, shp in sym.arg_spec.items():
            if k not in arg:
                raise KeyError(k + ' not initialized')
            if tuple(arg[k].shape) != tuple(shp):
                raise ValueError('shape inconsistent for %s inferred %s provided %s' % (k, shp, arg[k].shape))
        dev, f32 = self.device, torch.float32
        A = cfg.network.NUM_ANCHORS
        # heads: one GEMM for both RPN convs, one for both R-FCN convs
        std = np.tile(np.asarray(cfg.network.ANCHOR_STDS, np.float32), A)
        mean = np.tile(np.asarray(cfg.network.ANCHOR_MEANS, np.float32), A)
        wb, bb = arg['rpn_bbox_pred_weight'].reshape(4 * A, 512), arg['rpn_bbox_pred_bias']
        if cfg.network.NORMALIZE_RPN:   # rpn_inv_normalize folded: (Wx+b)*std+mean
            wb, bb = wb * std[:, None], bb * std + mean
        # both RPN convolutions as ONE 1x1 convolution of the own family on the NCHW map: output channels [score 2A | delta 4A] padded to the
        # 64-channel tile; three exact bf16 pieces in the fp32 mode (no scale to know: the feature comes out of the warp / aggregation
        # kernels), one piece in the bf16 mode
        w64, b64 = np.zeros((64, 512, 1, 1), np.float32), np.zeros(64, np.float32)
        w64[:6 * A, :, 0, 0] = np.concatenate([arg['rpn_cls_score_weight'].reshape(2 * A, 512), wb], 0)
        b64[:6 * A] = np.concatenate([arg['rpn_cls_score_bias'], bb], 0)
        self.rpn_sw = hip.SplitWeight(_t(w64, dev, f32), real_cout=6 * A, pieces=3 if self.pieces == 2 else self.pieces)
        self.rpn_b = _t(b64, dev, f32)
        self.n_cls_ch = arg['rfcn_cls_weight'].shape[0]
        # position-sensitive layout: GEMM [HW,512] x [512, 49*(ncls+nbox)], row (bin*D + d) of the permuted weight
        G = 7
        self.ncls = self.n_cls_ch // (G * G)
        self.nbox = arg['rfcn_bbox_weight'].shape[0] // (G * G)
        wc = arg['rfcn_cls_weight'].reshape(self.ncls, G * G, 512)
        wx = arg['rfcn_bbox_weight'].reshape(self.nbox, G * G, 512)
        w_ps = np.concatenate([wc, wx], 0).transpose(1, 0, 2).reshape(-1, 512)          # (49*D, 512)
        b_ps = np.concatenate([arg['rfcn_cls_bias'].reshape(self.ncls, G * G), arg['rfcn_bbox_bias'].reshape(self.nbox, G * G)], 0).T.reshape(-1)
        # as a 1x1 convolution of the own family: output channels padded to its 64-channel tiles (the padding columns are never read)
        self.ps_ld = -(-w_ps.shape[0] // 64) * 64
        wp, bp = _pad_rows(_t(w_ps.reshape(-1, 512, 1, 1), dev, f32), _t(b_ps, dev, f32), self.ps_ld)
        self.rfcn_sw, self.rfcn_b_ps = hip.SplitWeight(wp, real_cout=w_ps.shape[0], pieces=self.pieces), bp
        self.proposal = hip.ProposalOp(feature_stride=cfg.network.RPN_FEAT_STRIDE, scales=cfg.network.ANCHOR_SCALES,
                                       ratios=cfg.network.ANCHOR_RATIOS, rpn_pre_nms_top_n=cfg.TEST.RPN_PRE_NMS_TOP_N,
                                       rpn_post_nms_top_n=cfg.TEST.RPN_POST_NMS_TOP_N, threshold=cfg.TEST.RPN_NMS_THRESH,
                                       rpn_min_size=cfg.TEST.RPN_MIN_SIZE)
        two = self.pieces == 2
        self._slots = {'backbone': _Slots(128, dev, two), 'small': _Slots(16, dev, two), 'agg': _Slots(8, dev, two), 'flow': _Slots(16, dev, two),
                       'heads': _Slots(2, dev, two)}
        W = lambda name: _t(arg[name], dev, f32)
        if sym.kind in ('key', 'batch'):
            self.net = _ResNetWeights(arg, aux, '', 4, cfg.network.add_dcn, True, dev, self.pieces)
            self.feat_w, self.feat_b = hip.SplitWeight(W('feat_conv_3x3_weight'), pieces=self.pieces), W('feat_conv_3x3_bias')
            self._flownet_prepare({k: _t(v, dev, f32) for k, v in arg.items()
                                   if k.startswith(('flow_conv1', 'conv', 'Convolution', 'deconv', 'upsample_flow')) and 'stage' not in k
                                   and not k.startswith('conv0')})
            if sym.kind == 'batch':
                pass
            elif cfg.network.add_Nq_net:
                # Nq_net (:94-109): 3x3 1024 -> 256 + ReLU, 1x1 256 -> 16 + ReLU, 1x1 16 -> 1; the 16 / 1 output channels are padded to
                # the kernels' 64-channel tiles with zero weights
                w2, b2 = _pad_rows(W('Nq_conv2_weight'), W('Nq_conv2_bias'), 64)
                w3, b3 = _pad_rows(W('Nq_conv3_weight'), W('Nq_conv3_bias'), 64, 64)
                self.nq = [(hip.SplitWeight(W('Nq_conv1_weight'), pieces=self.pieces), W('Nq_conv1_bias')),
                           (hip.SplitWeight(w2, real_cout=16, pieces=self.pieces), b2),
                           (hip.SplitWeight(w3, real_cout=1, real_cin=16, pieces=self.pieces), b3)]
            elif cfg.network.add_Fgfa_net:
                self.em = [(hip.SplitWeight(W('em_conv%d_weight' % i), pieces=self.pieces), W('em_conv%d_bias' % i)) for i in (1, 2, 3)]
        else:
            self.rnet_w = _t(arg['rnet_conv0_weight'].reshape(1024, 3), dev, f32)
            self.rnet_b = W('rnet_conv0_bias')
            if cfg.network.add_small_net:
                self.small = _ResNetWeights(arg, aux, 'small_net_', 1, False, False, dev, self.pieces)
                self.fuse_w, self.fuse_b = hip.SplitWeight(W('fuse_reduce_add_weight'), pieces=self.pieces), W('fuse_reduce_add_bias')

    def _tap(self, name, x):
        if self.taps is not None:
            self.taps[name] = x

    def check_status(self):
        """Raises LsfaError if a convolution of this executor produced a non-finite value since the last check (lsfa_status_check:
        e.g. an under-estimated fp16 scale).  Synchronises the current stream: call it where the frame loop synchronises anyway."""
        hip.check_status(self.status)

    # ---- dense helpers ---------------------------------------------------------------
    def _conv(self, x, sw, bias=None, stride=1, pad=0, dil=1, act=0, amax_in=None, amax_out=None, **kw):
        """lsfa_conv_fwd on a channels-last map; the amax plumbing only exists for two-piece weights"""
        if sw.pieces != 2:
            amax_in = amax_out = None
        elif amax_in is None:
            amax_in = hip.amax_partial(x)
        return hip.conv_split(x, sw, bias, stride, pad, dil, act=act, amax_in=amax_in, amax_out=amax_out, status=self.status, **kw)

    def _resnet(self, x, net, stages, section):
        """Pre-activation ResNet (resnet.py:138-240) on channels-last activations, every contraction on lsfa_conv_fwd.
        x (N, 3, H, W) NCHW -> (map (N, h, w, C) channels-last, its amax slots): relu(bn1(.)) of the last unit's output when the
        network has a tail (the backbone), else the last unit's raw output (the small net's stage 1, `need_part`)."""
        S = self._slots[section].begin()
        units = [u for u in net.units if u['stage'] <= stages]
        # bn_data + conv0 + bn0 + relu0, then pool0 with the first unit's bn1 + relu1 as a second output: two launches
        if net.conv0_exact is not None:
            if isinstance(x, hip.ImageTable):
                x = x.materialize()
            xa = (x * net.bn_data[0].view(1, 3, 1, 1) + net.bn_data[1].view(1, 3, 1, 1)).permute(0, 2, 3, 1)
            xh = torch.zeros(tuple(xa.shape[:3]) + (32,), device=x.device, dtype=torch.float32)
            xh[..., :3] = xa
            y = hip.conv_split(xh, net.conv0_exact, net.conv0_b, 2, 3, 1, act=1)
        else:
            y = hip.stem_conv(x, net.conv0_w_l, net.conv0_b, net.bn_data[0], net.bn_data[1])
        am_a = S.new()             # pool0 publishes the maximum of its second output like the convolutions' epilogues do
        x4, a = hip.maxpool3x3s2_nhwc(y, scale2=units[0]['bn1'][0], shift2=units[0]['bn1'][1], amax_out=am_a)
        dilate = 1
        pre = None                 # (scale, shift) of this unit's bn1 when its input `a` is still the raw sum (applied where conv1 cuts it)
        for ui, u in enumerate(units):
            first = u['unit'] == 1
            # stage 4 keeps stride 1 and doubles the dilation from its 2nd unit on (resnet.py:33-34, :72-76, :223-230)
            stride = 2 if (first and u['stage'] in (2, 3)) else 1
            ud = dilate
            if first and u['stage'] == 4:
                dilate = dilate * 2
            am_c1, am_c2, am_n = S.new(), S.new(), S.new()
            if pre is None:
                c1 = self._conv(a, u['w1'], u['b1'], act=1, amax_in=am_a, amax_out=am_c1)     # conv1 + folded bn2 + relu2
            else:                  # ... on max(sum * bn1 scale + bn1 shift, 0), which only this convolution reads: never stored
                c1 = self._conv(a, u['w1'], u['b1'], act=1, amax_in=am_a, amax_out=am_c1, in_scale=pre[0], in_shift=pre[1])
            if u['dcn']:
                # DeformableConvolution (sym_common.py:138-157): offsets, bilinear columns, contraction + folded bn3 + relu3.  Every column
                # entry is an interpolation of c1 (zeros outside): max|col| <= max|c1|, so c1's scale serves the contraction
                off = self._conv(c1, u['off'], u['off_b'], 1, ud, ud, amax_in=am_c1)
                col = hip.deform_im2col_cl(c1, off, 3, 3, ud, 1, ud, P.NUM_DEFORMABLE_GROUP)
                c2 = self._conv(col.view(c1.shape[0], c1.shape[1], c1.shape[2], -1), u['w2'], u['b2'], act=1, amax_in=am_c1, amax_out=am_c2)
            else:
                c2 = None
            # shortcut: 1x1 (stride) on relu1, or x
            if first and pre is not None:      # the strided shortcut reads the same raw sum through the same bn1 + relu1
                sc = self._conv(a, u['sc'], None, stride, amax_in=am_a, in_scale=pre[0], in_shift=pre[1])
            else:
                sc = self._conv(a, u['sc'], None, stride, amax_in=am_a) if first else x4
            if c2 is None:
                c2 = self._conv(c1, u['w2'], u['b2'], stride, ud, ud, act=1, amax_in=am_c1, amax_out=am_c2)   # conv2 + folded bn3 + relu3
            # conv3 + shortcut add in place + the bn1 / relu1 the NEXT unit (or the tail) applies to the sum, as a second output
            nxt = units[ui + 1]['bn1'] if ui + 1 < len(units) else net.bn1
            # The tail feeds relu1 to a padded 3x3 (its zeros are not max(0 * s + t, 0)) and pool0 produces the first one: there the
            # activated map is stored.  Everywhere else (32 of ResNet-101's 33 units) its readers are conv1 and, in a stage's first
            # unit, the shortcut's strided 1x1, which apply bn1 + relu1 themselves: conv3 then writes the sum alone and only publishes max(relu1) for conv1's fp16 scale - a quarter less
            # map traffic per unit (the sum read and written, the activated map written and read, were its four big streams).
            lone = self.input_activation_at_cut and ui + 1 < len(units) and u['w1'].pieces != 3 and \
                sc.numel() * 4 >= self.input_activation_min_bytes
            pre = None
            if nxt is not None and lone:
                if u['w3'].pieces == 2:      # max(relu1) is the next conv1's fp16 scale; the one-piece (bf16) mode has no scale
                    x4 = self._conv(c2, u['w3'], None, amax_in=am_c2, amax_out=am_n, out=sc, residual=sc, scale2=nxt[0], shift2=nxt[1])
                else:
                    x4 = self._conv(c2, u['w3'], None, out=sc, residual=sc)
                a, pre = x4, nxt
            elif nxt is not None:
                x4, a = self._conv(c2, u['w3'], None, amax_in=am_c2, amax_out=am_n, out=sc, residual=sc, out2=torch.empty_like(sc),
                                   scale2=nxt[0], shift2=nxt[1])
            else:
                x4, a = self._conv(c2, u['w3'], None, amax_in=am_c2, amax_out=am_n, out=sc, residual=sc), None
            am_a = am_n
        return (a if a is not None else x4), am_a

    def _backbone(self, data):
        """ResNet-101 (+ DCN) + feat_conv_3x3 (3x3 pad 6 dilate 6, 2048 -> 1024, bias, ReLU; :52-54) -> NCHW, what the warp /
        aggregation kernels and the API take"""
        a, am = self._resnet(data, self.net, 4, 'backbone')
        return self._conv(a, self.feat_w, self.feat_b, 1, 6, 6, act=1, amax_in=am, nchw=True)

    def _flownet_prepare(self, fw):
        """Weights of FlowNet-S: fragments for the convolutions and the four phases of each Deconvolution(4x4, stride 2) + Crop(1)
        (input channels zero-padded to the concatenated maps' padded widths)."""
        dev, pc = self.device, self.flow_pieces
        own = {}
        w1 = fw['flow_conv1_weight']                                   # (64, 6, 7, 7)
        own['c1_cur'] = hip.stem_weight_layout(w1[:, 0:3].contiguous())
        own['c1_ref'] = hip.stem_weight_layout(w1[:, 3:6].contiguous())
        own['in_scale'] = torch.full((3,), 1.0 / 255.0, device=dev)
        own['in_shift'] = torch.zeros(3, device=dev)
        for name in ('conv2', 'conv3', 'conv3_1', 'conv4', 'conv4_1', 'conv5', 'conv5_1', 'conv6', 'conv6_1'):
            own[name] = hip.SplitWeight(fw[name + '_weight'], pieces=pc)

        def pad32(c):
            return -(-c // 32) * 32
        for name in ('deconv5', 'deconv4', 'deconv3', 'deconv2'):
            wt = fw[name + '_weight']                                     # (Cin, Cout, 4, 4)
            own[name] = hip.deconv_phase_weights(wt, cin_pad=pad32(wt.shape[0]), pieces=pc)
        ws = fw['Convolution5_scale_weight']                              # (1024, 194, 1, 1)
        wsp, _ = _pad_rows(ws, None, ws.shape[0], pad32(ws.shape[1]))
        own['scale'] = hip.SplitWeight(wsp, real_cin=ws.shape[1], pieces=pc)
        for name in ('Convolution1', 'Convolution2', 'Convolution3', 'Convolution4', 'Convolution5'):
            own[name] = fw[name + '_weight'].permute(0, 2, 3, 1).contiguous()      # (2, 3, 3, Cin)
        self._flow_own, self.flow = own, fw

    def _flownet(self, img_cur, img_ref):
        """FlowNet-S (:150-207) on the half-resolution pair, channels-last, no library call.  Concat nodes are channel slices of maps the
        producers write into directly (operand views of lsfa_conv_fwd), Deconvolution + Crop are four phase convolutions in one launch,
        the 2-channel heads and their upsampling are small dedicated kernels (csrc/flownet.hip).  -> flow (N,2,h,w), scale map (N,1024,h,w)"""
        o, fw, dev = self._flow_own, self.flow, self.device
        LEAKY = 2
        N = img_cur.shape[0]
        st = self.status
        S = self._slots['flow'].begin()

        def cmap(h, w, c):      # a concatenated map with its channel count padded to a multiple of 32 (the padding stays zero)
            return hip.zeros_f32((N, h, w, -(-c // 32) * 32), dev)      # (the padding channels are multiplied by zero weights: they must be finite)

        def out_hw(h, w, k, stride, pad):
            return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1

        def conv(x, am_x, name, stride, pad, out=None, am_out=None, cin=None):
            """-> (out, its amax slots); a Concat map (`out` given) collects the maxima of all its producers in ONE slot row"""
            sw = o[name]
            h, w = out_hw(x.shape[1], x.shape[2], sw.kh, stride, pad)
            if out is None:
                out, am_out = torch.empty((N, h, w, sw.cout), device=dev, dtype=torch.float32), S.new()
            hip.conv_split_view(x, sw, fw[name + '_bias'], out, stride=stride, pad=(pad, pad), act=LEAKY, cin=cin, amax_in=am_x,
                                amax_out=am_out, status=st)
            return out, am_out

        def deconv(x, am_x, name, out, am_out, c0):
            hip.deconv4x4s2_crop(x, o[name], fw[name + '_bias'], out, c0=c0, act=LEAKY, amax_in=am_x, amax_out=am_out, status=st)

        def head(x, name, cin):
            return hip.head_conv3x3(x, o[name], fw[name + '_bias'], cin=cin)

        def upflow(f, name, out, am_out, c0):
            hip.upsample_flow(f, fw[name + '_weight'], fw[name + '_bias'], out, c0, amax_out=am_out)

        # avg pool 2x2 of each image (x 1/255 folded into the first convolution's input affine), flow_conv1 as two passes
        pc, pr = hip.avgpool_nchw(img_cur, 2), hip.avgpool_nchw(img_ref.contiguous(), 2)
        r1 = hip.stem_conv(pc, o['c1_cur'], None, o['in_scale'], o['in_shift'], act=0)
        am_r1 = S.new()
        r1 = hip.stem_conv(pr, o['c1_ref'], fw['flow_conv1_bias'], o['in_scale'], o['in_shift'], out=r1, accum=r1, act=LEAKY, amax_out=am_r1)
        h2, w2 = out_hw(r1.shape[1], r1.shape[2], 5, 2, 2)
        c5, am5 = cmap(h2, w2, 194), S.new()
        conv(r1, am_r1, 'conv2', 2, 2, out=c5, am_out=am5)                # r2 = c5[..., :128]
        r3, am_r3 = conv(c5, am5, 'conv3', 2, 2, cin=128)
        c4, am4 = cmap(r3.shape[1], r3.shape[2], 386), S.new()
        conv(r3, am_r3, 'conv3_1', 1, 1, out=c4, am_out=am4)              # r4 = c4[..., :256]
        r5, am_r5 = conv(c4, am4, 'conv4', 2, 1, cin=256)
        c3, am3 = cmap(r5.shape[1], r5.shape[2], 770), S.new()
        conv(r5, am_r5, 'conv4_1', 1, 1, out=c3, am_out=am3)              # r6 = c3[..., :512]
        r7, am_r7 = conv(c3, am3, 'conv5', 2, 1, cin=512)
        c2, am2 = cmap(r7.shape[1], r7.shape[2], 1026), S.new()
        conv(r7, am_r7, 'conv5_1', 1, 1, out=c2, am_out=am2)              # r8 = c2[..., :512]
        r9, am_r9 = conv(c2, am2, 'conv6', 2, 1, cin=512)
        r10, am_r10 = conv(r9, am_r9, 'conv6_1', 1, 1)
        f6 = head(r10, 'Convolution1', 1024)
        deconv(r10, am_r10, 'deconv5', c2, am2, 512)
        upflow(f6, 'upsample_flow6to5', c2, am2, 1024)
        f5 = head(c2, 'Convolution2', 1026)
        deconv(c2, am2, 'deconv4', c3, am3, 512)
        upflow(f5, 'upsample_flow5to4', c3, am3, 768)
        f4 = head(c3, 'Convolution3', 770)
        deconv(c3, am3, 'deconv3', c4, am4, 256)
        upflow(f4, 'upsample_flow4to3', c4, am4, 384)
        f3 = head(c4, 'Convolution4', 386)
        deconv(c4, am4, 'deconv2', c5, am5, 128)
        upflow(f3, 'upsample_flow3to2', c5, am5, 192)
        c5p = hip.avgpool2_nhwc(c5)                                       # an average never exceeds its inputs: c5's slots bound c5p
        flow = hip.head_conv3x3(c5p, o['Convolution5'], fw['Convolution5_bias'], cin=194, mul=2.5, nchw=True)
        scale = self._conv(c5p, o['scale'], fw['Convolution5_scale_bias'], amax_in=am5, nchw=True)
        return flow, scale

    def _heads(self, conv_feat, im_info):
        """SliceChannel -> RPN -> Proposal -> R-FCN maps -> PSROI + average + softmax (:479-546)."""
        cfg = self.cfg
        A = cfg.network.NUM_ANCHORS
        n, _, h, w = conv_feat.shape
        # both RPN convolutions on the matrix pipe straight from the NCHW map (the direct kernel's K-major operand form), then the per-anchor
        # softmax + the split into MultiProposal's two NCHW inputs
        logits = hip.conv_split(conv_feat, self.rpn_sw, self.rpn_b, x_nchw=True, status=self.status)
        cls_prob, rpn_bbox = hip.rpn_softmax_split(logits, A)
        rois = self.proposal(cls_prob, rpn_bbox, im_info)
        D = self.ncls + self.nbox
        # both R-FCN convolutions as ONE 1x1 convolution of the own family that writes the position-sensitive layout [h][w][bin][class | box]
        # directly: channels 512.. of the NCHW feature turned channels-last (one copy that also leaves their maximum), then lsfa_conv_fwd (cells
        # self.ps_ld floats apart)
        S = self._slots['heads'].begin()
        am = S.new()
        rows = hip.nchw_to_nhwc(conv_feat, 512, 512, amax_out=am)
        ps = self._conv(rows, self.rfcn_sw, self.rfcn_b_ps, amax_in=am)
        if self.taps is not None:
            nchw = ps[..., :49 * D].reshape(n, h * w, 49, D).permute(0, 3, 2, 1).reshape(n, D * 49, h, w)
            self.taps.update(rpn_cls_prob=cls_prob, rpn_bbox_pred=rpn_bbox, cls_map=nchw[:, :self.n_cls_ch],
                             box_map=nchw[:, self.n_cls_ch:])
        cls_p, bbox = hip.rfcn_head_ps_ld(ps, self.ps_ld, rois, h, w, self.ncls, self.nbox, 0.0625, 7, 7)
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

    def _pair_rows(self, first, second, amax_out):
        """Concat(first, second) on the batch axis (:95, :133) as channels-last images: (2N, H, W, C); their maximum into amax_out"""
        n, c, h, w = first.shape
        x = torch.empty((2 * n, h, w, c), device=first.device, dtype=torch.float32)
        hip.nchw_to_nhwc(first, out=x[:n], amax_out=amax_out)
        hip.nchw_to_nhwc(second, out=x[n:], amax_out=amax_out)
        return x

    def _key_aggregate(self, conv_feat, flow, scale_map, feat_key_old):
        cfg = self.cfg
        if flow is not None:
            warp = hip.warp_bilinear(feat_key_old, flow, mul=scale_map)
            if self.taps is not None:
                self.taps.update(flow=flow, scale_map=scale_map, warp=warp)
            n, _, h, w = warp.shape
            if cfg.network.add_Nq_net:
                # Nq_net (:94-109) on Concat(warp, conv_feat): rows 0..n-1 weight the warped maps, n..2n-1 the current ones
                S = self._slots['agg'].begin()
                am0, am1, am2 = S.new(), S.new(), S.new()
                x = self._pair_rows(warp, conv_feat, am0)
                x = self._conv(x, self.nq[0][0], self.nq[0][1], 1, 1, 1, act=1, amax_in=am0, amax_out=am1)
                x = self._conv(x, self.nq[1][0], self.nq[1][1], act=1, amax_in=am1, amax_out=am2)
                # the last convolution (1 output channel, padded to 64) written NCHW: channel 0 of image r is logit row r, read in place
                x = self._conv(x, self.nq[2][0], self.nq[2][1], amax_in=am2, nchw=True)
                if self.taps is not None:
                    self._tap('nq_logits', x[:, 0:1].contiguous())
                conv_feat = hip.aggregate_softmax2(warp, conv_feat, x, logit_row_stride=x.shape[1] * h * w)
            elif cfg.network.add_Fgfa_net:
                # get_embednet on Concat(conv_feat, warp) (:118-135; note the order, :133): 1x1 1024 -> 512, 3x3 512 -> 512, 1x1 512 -> 2048
                S = self._slots['agg'].begin()
                am0, am1, am2 = S.new(), S.new(), S.new()
                x = self._pair_rows(conv_feat, warp, am0)
                x = self._conv(x, self.em[0][0], self.em[0][1], act=1, amax_in=am0, amax_out=am1)
                x = self._conv(x, self.em[1][0], self.em[1][1], 1, 1, 1, act=1, amax_in=am1, amax_out=am2)
                e = self._conv(x, self.em[2][0], self.em[2][1], amax_in=am2, nchw=True)
                self._tap('embed', e)
                conv_feat = hip.aggregate_cosine(warp, conv_feat, e[n:2 * n], e[0:n])
            else:
                conv_feat = 0.5 * (warp + conv_feat)
        return conv_feat

    def _key_heads(self, conv_feat, im_info):
        rois, cls_prob, bbox_pred = self._heads(conv_feat, im_info)
        return {'choose_feat_output': conv_feat, 'rois_output': rois, 'cls_prob_reshape_output': cls_prob,
                'bbox_pred_reshape_output': bbox_pred}

    def small_net_feature(self, data, nchw=True):
        """fuse_small_net's image branch (:209-236): avgpool 4x4 -> small_net_ stem + stage 1 ->
        fuse_reduce_add (3x3 256 -> 1024, written NCHW: the warp kernel's `add` operand; nchw=False: channels-last, the operand of
        lsfa_warp_bilinear_cl).  It depends on the frame image only, so a caller may compute it ahead of the rest of the frame
        (lsfa_amd/core/graphs.py overlaps it with the previous frame's tail)."""
        with torch.no_grad():
            img = hip.avgpool_nchw(data, 4)
            s, am = self._resnet(img, self.small, 1, 'small')
            return self._conv(s, self.fuse_w, self.fuse_b, 1, 1, 1, amax_in=am, nchw=nchw)

    def cur_channels_last(self, channels):
        """whether a non-key frame whose small-net feature is computed inside forward() runs on channels-last maps (_forward_cur_cl)"""
        return bool(self.cfg.network.add_small_net and CUR_CHANNELS_LAST and self.pieces != 0 and channels % 4 == 0)

    def _forward_cur_cl(self, d):
        """The non-key frame on channels-last maps (r6): the warped feature is read by two 1x1 convolutions only - GEMMs over the channel axis -
        so it is produced as (pixel, channel) rows: the key feature turned channels-last once per PASS (not half a map per frame in front of
        the R-FCN convolution), the small net's fuse convolution in its natural layout, lsfa_warp_bilinear_cl (the same bits as the NCHW
        kernel, + the maximum the R-FCN convolution's scale needs).  `conv_feat` in the outputs is the NCHW view of the same memory."""
        S = self._slots['heads'].begin()
        feat_cl = d.get('feat_key_cl')                 # (B, H, W, C): a caller that hands the key feature over channels-last (FramePipeline does, as its
        if feat_cl is None:                            #  hand-over copy) saves the per-pass transposition
            feat_cl = hip.nchw_to_nhwc(d['feat_key'])
        add_cl = self.small_net_feature(d['data'], nchw=False)
        self._tap('small_feat', add_cl.permute(0, 3, 1, 2))
        am = S.new()
        conv_cl = hip.warp_bilinear_cl(feat_cl, d['motion_vector'], add_cl=add_cl, res=d['res_diff'], res_w=self.rnet_w, res_b=self.rnet_b,
                                       amax_out=am, amax_c0=512)     # the maximum of the channels the R-FCN convolution (two fp16 pieces) reads
        rois, cls_prob, bbox_pred = self._heads_cl(conv_cl, am, d['im_info'])
        return {'data': d['data'], 'data_key': d.get('data_key'), 'data_key_old': d.get('data_key_old'),
                'feat_key_old': d.get('feat_key_old'), 'rois_output': rois, 'cls_prob_reshape_output': cls_prob,
                'bbox_pred_reshape_output': bbox_pred, 'conv_feat': conv_cl.permute(0, 3, 1, 2)}

    def _heads_cl(self, conv_cl, am, im_info):
        """_heads on a channels-last feature (N, H, W, 1024) whose maximum sits in `am`: both convolutions read their 512 channels in place."""
        cfg = self.cfg
        A = cfg.network.NUM_ANCHORS
        n, h, w, _ = conv_cl.shape
        logits = torch.empty((n, h, w, self.rpn_sw.cout), device=conv_cl.device, dtype=torch.float32)
        hip.conv_split_view(conv_cl, self.rpn_sw, self.rpn_b, logits, cin=512, status=self.status)      # (three exact bf16 pieces, or one in bf16 mode: no scale)
        cls_prob, rpn_bbox = hip.rpn_softmax_split(logits, A)
        rois = self.proposal(cls_prob, rpn_bbox, im_info)
        D = self.ncls + self.nbox
        ps = torch.empty((n, h, w, self.rfcn_sw.cout), device=conv_cl.device, dtype=torch.float32)
        hip.conv_split_view(conv_cl, self.rfcn_sw, self.rfcn_b_ps, ps, cin=512, cin0=512, amax_in=am, status=self.status)
        if self.taps is not None:
            nchw = ps[..., :49 * D].reshape(n, h * w, 49, D).permute(0, 3, 2, 1).reshape(n, D * 49, h, w)
            self.taps.update(rpn_cls_prob=cls_prob, rpn_bbox_pred=rpn_bbox, cls_map=nchw[:, :self.n_cls_ch],
                             box_map=nchw[:, self.n_cls_ch:])
        cls_p, bbox = hip.rfcn_head_ps_ld(ps, self.ps_ld, rois, h, w, self.ncls, self.nbox, 0.0625, 7, 7)
        B = cfg.TEST.BATCH_IMAGES
        return rois, cls_p.view(B, -1, cls_p.shape[1]), bbox.view(B, -1, bbox.shape[1])

    def _forward_cur(self, d):
        cfg = self.cfg
        add = d.get('small_feat')          # precomputed by the caller, else computed here
        # r6: the whole non-key frame on channels-last maps (no transposing copy in front of the R-FCN convolution) whenever nothing forces the
        # operator layout: the small net's feature is computed here (not handed over NCHW), C is a multiple of 4, no exact-fp32 reference mode.
        # LSFA_CUR_NCHW=1 keeps the NCHW form (A/B; it is also what the key frames and the batch test symbol run).
        if add is None and self.cur_channels_last(d['feat_key'].shape[1]):
            return self._forward_cur_cl(d)
        if add is None and cfg.network.add_small_net:
            add = self.small_net_feature(d['data'])
        self._tap('small_feat', add)
        conv_feat = hip.warp_bilinear(d['feat_key'], d['motion_vector'], add=add, res=d['res_diff'], res_w=self.rnet_w,
                                      res_b=self.rnet_b)
        rois, cls_prob, bbox_pred = self._heads(conv_feat, d['im_info'])
        return {'data': d['data'], 'data_key': d.get('data_key'), 'data_key_old': d.get('data_key_old'),
                'feat_key_old': d.get('feat_key_old'), 'rois_output': rois, 'cls_prob_reshape_output': cls_prob,
                'bbox_pred_reshape_output': bbox_pred, 'conv_feat': conv_feat}
