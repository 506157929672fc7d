"""Parameter names and shapes of the LSFA test symbols, and seeded random initialisation.

Names are the MXNet layer names of the reference (SURVEY.md §8b "Symbol API"):
backbone   dff_rfcn/symbols/resnet.py:138-240 + sym_common.py:92-135, 249-290
symbols    dff_rfcn/symbols/resnet_v1_101_flownet_rfcn.py:44-236 (helpers), :448-659 (test symbols)
init       init_weight, :753-870 (normal(0, 0.01) for the new layers, zeros for DCN offsets,
           Convolution5_scale weight 0 / bias 1, small_net_* copied from the big net)
`arg` holds weights/biases/gamma/beta, `aux` holds moving_mean/moving_var, like MXNet.
"""
from collections import OrderedDict

import numpy as np

UNITS = (3, 4, 23, 3)
FILTERS = (256, 512, 1024, 2048)
DEFORMABLE_UNITS = (0, 1, 1, 3)   # resnet_v1_101_flownet_rfcn.py:45
NUM_DEFORMABLE_GROUP = 4          # :46


def is_dcn_unit(stage, unit, add_dcn=True):
    """resnet.py:166-230: unit i of stage s is deformable iff i >= units[s] - deformable_units[s] + 1."""
    return add_dcn and unit >= UNITS[stage - 1] - DEFORMABLE_UNITS[stage - 1] + 1


def _bn(arg, aux, name, c):
    arg[name + '_gamma'] = (c,)
    arg[name + '_beta'] = (c,)
    aux[name + '_moving_mean'] = (c,)
    aux[name + '_moving_var'] = (c,)


def resnet_spec(arg, aux, prefix='', stages=4, add_dcn=True, tail=True):
    _bn(arg, aux, prefix + 'bn_data', 3)
    arg[prefix + 'conv0_weight'] = (64, 3, 7, 7)
    _bn(arg, aux, prefix + 'bn0', 64)
    cin = 64
    for s in range(1, stages + 1):
        nf = FILTERS[s - 1]
        mid = nf // 4
        for u in range(1, UNITS[s - 1] + 1):
            p = '%sstage%d_unit%d_' % (prefix, s, u)
            _bn(arg, aux, p + 'bn1', cin)
            arg[p + 'conv1_weight'] = (mid, cin, 1, 1)
            _bn(arg, aux, p + 'bn2', mid)
            if is_dcn_unit(s, u, add_dcn):
                arg[p + 'conv2_offset_weight'] = (2 * 9 * NUM_DEFORMABLE_GROUP, mid, 3, 3)
                arg[p + 'conv2_offset_bias'] = (2 * 9 * NUM_DEFORMABLE_GROUP,)
            arg[p + 'conv2_weight'] = (mid, mid, 3, 3)
            _bn(arg, aux, p + 'bn3', mid)
            arg[p + 'conv3_weight'] = (nf, mid, 1, 1)
            if u == 1:
                arg[p + 'sc_weight'] = (nf, cin, 1, 1)
            cin = nf
    if tail:
        _bn(arg, aux, prefix + 'bn1', cin)


def _conv(arg, name, cout, cin, k, bias=True):
    arg[name + '_weight'] = (cout, cin, k, k)
    if bias:
        arg[name + '_bias'] = (cout,)


def flownet_spec(arg):
    for name, cout, cin, k in (('flow_conv1', 64, 6, 7), ('conv2', 128, 64, 5), ('conv3', 256, 128, 5),
                               ('conv3_1', 256, 256, 3), ('conv4', 512, 256, 3), ('conv4_1', 512, 512, 3),
                               ('conv5', 512, 512, 3), ('conv5_1', 512, 512, 3), ('conv6', 1024, 512, 3),
                               ('conv6_1', 1024, 1024, 3), ('Convolution1', 2, 1024, 3), ('Convolution2', 2, 1026, 3),
                               ('Convolution3', 2, 770, 3), ('Convolution4', 2, 386, 3), ('Convolution5', 2, 194, 3),
                               ('Convolution5_scale', 1024, 194, 1)):
        _conv(arg, name, cout, cin, k)
    # Deconvolution weights are (in, out, k, k)
    for name, cin, cout in (('deconv5', 1024, 512), ('deconv4', 1026, 256), ('deconv3', 770, 128), ('deconv2', 386, 64),
                            ('upsample_flow6to5', 2, 2), ('upsample_flow5to4', 2, 2), ('upsample_flow4to3', 2, 2),
                            ('upsample_flow3to2', 2, 2)):
        arg[name + '_weight'] = (cin, cout, 4, 4)
        arg[name + '_bias'] = (cout,)


def head_spec(arg, cfg):
    A = cfg.network.NUM_ANCHORS
    ncls = cfg.dataset.NUM_CLASSES
    nreg = 2 if cfg.CLASS_AGNOSTIC else ncls
    _conv(arg, 'rpn_cls_score', 2 * A, 512, 1)
    _conv(arg, 'rpn_bbox_pred', 4 * A, 512, 1)
    _conv(arg, 'rfcn_cls', 7 * 7 * ncls, 512, 1)
    _conv(arg, 'rfcn_bbox', 7 * 7 * 4 * nreg, 512, 1)


def key_symbol_spec(cfg):
    arg, aux = OrderedDict(), OrderedDict()
    resnet_spec(arg, aux, '', add_dcn=cfg.network.add_dcn)
    _conv(arg, 'feat_conv_3x3', 1024, 2048, 3)
    flownet_spec(arg)
    if cfg.network.add_Nq_net:
        _conv(arg, 'Nq_conv1', 256, 1024, 3)
        _conv(arg, 'Nq_conv2', 16, 256, 1)
        _conv(arg, 'Nq_conv3', 1, 16, 1)
    elif cfg.network.add_Fgfa_net:
        _conv(arg, 'em_conv1', 512, 1024, 1)
        _conv(arg, 'em_conv2', 512, 512, 3)
        _conv(arg, 'em_conv3', 2048, 512, 1)
    head_spec(arg, cfg)
    return arg, aux


def batch_symbol_spec(cfg):
    """get_batch_test_symbol (:661-751): backbone + FlowNet + heads, no aggregation nets."""
    arg, aux = OrderedDict(), OrderedDict()
    resnet_spec(arg, aux, '', add_dcn=cfg.network.add_dcn)
    _conv(arg, 'feat_conv_3x3', 1024, 2048, 3)
    flownet_spec(arg)
    head_spec(arg, cfg)
    return arg, aux


def cur_symbol_spec(cfg):
    arg, aux = OrderedDict(), OrderedDict()
    if cfg.network.rnet_num_conv != 0 or cfg.network.res_diff_bn or cfg.network.fuse_type != 'add' \
            or 'conv' in str(cfg.network.fnet_type):
        raise NotImplementedError("only the trained LSFA configuration (rnet_num_conv 0, fuse 'add', no fnet) is built")
    _conv(arg, 'rnet_conv0', 1024, 3, 1)
    if cfg.network.add_small_net:
        if cfg.network.small_net_stride != 4 or cfg.network.small_net_fuse_type != 'add' \
                or cfg.network.small_net_bn_before_fuse or cfg.network.small_net_scale_before_fuse:
            raise NotImplementedError("only small_net_stride 4 / fuse 'add' is built")
        resnet_spec(arg, aux, 'small_net_', stages=1, add_dcn=False, tail=False)
        _conv(arg, 'fuse_reduce_add', 1024, 256, 3)
    head_spec(arg, cfg)
    return arg, aux


def init_params(cfg, seed=0, head_fg_prior=0.02, dcn_offset_std=0.02):
    """Seeded random weights for both symbols as numpy float32 dicts (arg_params, aux_params).

    SURVEY.md §8d: N(0, 0.01) for the layers init_weight creates, He-normal for the backbone and
    FlowNet convolutions (with the last conv of every residual branch scaled down so activations stay
    O(1) through 33 units), BN gamma 1 / beta 0 / mean 0 / var 1 except bn_data which carries pixel
    statistics, Convolution5_scale weight 0 / bias 1 (:869-870), RPN foreground bias shifted so that
    about `head_fg_prior` of the anchors score above 0.5.  The reference initialises the DCN offset
    branches to zero (sym_common.py:250-256); `dcn_offset_std` > 0 perturbs them so the bilinear
    sampling path does real work in benchmarks (0 restores the reference init).
    """
    rs = np.random.RandomState(seed)
    karg, kaux = key_symbol_spec(cfg)
    carg, caux = cur_symbol_spec(cfg)
    arg, aux = OrderedDict(), OrderedDict()

    def he(shape, gain=1.0):
        fan_in = float(np.prod(shape[1:]))
        return (rs.randn(*shape) * (gain * np.sqrt(2.0 / fan_in))).astype(np.float32)

    new_layers = ('feat_conv_3x3', 'rpn_', 'rfcn_', 'Nq_', 'em_', 'rnet_', 'fuse_reduce_add')
    for name, shape in list(karg.items()) + list(carg.items()):
        if name in arg:
            continue
        if name.startswith('small_net_'):
            continue  # copied below (init_weight :755-760)
        if name.endswith('_gamma'):
            v = np.ones(shape, np.float32)
        elif name.endswith('_beta'):
            v = np.zeros(shape, np.float32)
        elif 'offset' in name:
            v = (rs.randn(*shape) * dcn_offset_std).astype(np.float32) if name.endswith('weight') else np.zeros(shape, np.float32)
        elif name == 'Convolution5_scale_weight':
            v = np.zeros(shape, np.float32)
        elif name == 'Convolution5_scale_bias':
            v = np.ones(shape, np.float32)
        elif name.endswith('_bias'):
            v = np.zeros(shape, np.float32)
        elif name.startswith(new_layers):
            std = 0.01
            if name.startswith('feat_conv_3x3'):
                v = he(shape)            # keeps the 1024-d feature O(1) under random backbone weights
            else:
                v = (rs.randn(*shape) * std).astype(np.float32)
        elif name.startswith(('deconv', 'upsample_flow')):
            fan_in = float(shape[0] * 4)  # each output pixel sees (k/stride)^2 = 4 taps per input channel
            v = (rs.randn(*shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        elif name.startswith('Convolution'):
            v = he(shape, 0.1)           # flow heads: small flows (cells)
        elif name.endswith('conv3_weight'):
            v = he(shape, 0.25)          # damp every residual branch
        else:
            v = he(shape)
        arg[name] = v
    for name, shape in list(kaux.items()) + list(caux.items()):
        if name.startswith('small_net_'):
            continue
        aux[name] = np.zeros(shape, np.float32) if name.endswith('moving_mean') else np.ones(shape, np.float32)
    aux['bn_data_moving_mean'] = np.array([110.0, 115.0, 120.0], np.float32)
    aux['bn_data_moving_var'] = np.array([60.0 ** 2, 60.0 ** 2, 60.0 ** 2], np.float32)
    # class / fg priors
    A = cfg.network.NUM_ANCHORS
    b = arg['rpn_cls_score_bias']
    b[A:] = np.log(head_fg_prior / (1 - head_fg_prior))
    arg['rfcn_cls_weight'] *= 8.0        # spread the class scores so detections are not uniform
    arg['rfcn_bbox_weight'] *= 2.0
    arg['rfcn_cls_bias'][:49] = 1.0      # background prior
    # small net seeded from the big net (init_weight :755-760)
    for name in list(carg.keys()):
        if name.startswith('small_net_'):
            arg[name] = arg[name.replace('small_net_', '')].copy()
    for name in list(caux.keys()):
        if name.startswith('small_net_'):
            aux[name] = aux[name.replace('small_net_', '')].copy()
    return arg, aux
