"""torch-CPU statement of LSFA's key / cur test graphs — TEST INFRASTRUCTURE ONLY.

Follows the reference's symbol definition layer by layer, UNFUSED (every BatchNorm is its
own op, every 1x1 convolution is a convolution, rpn_inv_normalize is applied after the conv):
  dff_rfcn/symbols/resnet.py:70-101, :138-240           pre-activation ResNet-101
  dff_rfcn/symbols/sym_common.py:92-157, :249-290        bn / conv / deformable conv
  dff_rfcn/symbols/resnet_v1_101_flownet_rfcn.py:44-236  feat conv, rnet, Nq, FlowNet, small net
  :448-551 key symbol, :553-659 cur symbol
The dense ops are torch.nn.functional on the CPU in fp32; the custom stages (warp, aggregate,
Proposal, PSROI, DCN im2col) are the C oracle.  PARITY UNPINNED for all of it: MXNet is not
vendored and the reference holds no golden outputs for these graphs.

`dtype=torch.float64` (r4) runs every DENSE op (convolutions, BatchNorm, pooling, softmax, the DCN contraction) in
float64 and hands float32-rounded maps to the custom stages at exactly the boundaries where the reference's operators
receive fp32 tensors.  That graph is the yardstick of tests/test_parity_fullres_gpu.py: the fp32 statement of this file
and the GPU path are two fp32 evaluations of the same graph with different summation orders; how far each is from the
float64 graph says whether the GPU path is "no worse than the reference's own fp32" without appealing to a constant.
"""
import numpy as np
import torch
import torch.nn.functional as F

import oracle

EPS = 2e-5
UNITS = (3, 4, 23, 3)
FILTERS = (256, 512, 1024, 2048)
DEFORMABLE_UNITS = (0, 1, 1, 3)


def _T(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


class Params(object):
    def __init__(self, arg, aux, dtype=torch.float32):
        self.arg, self.aux, self.dt = arg, aux, dtype

    def T(self, a):
        """numpy (or tensor) -> tensor of the graph's dense dtype; fp32 values are represented exactly in float64"""
        if isinstance(a, torch.Tensor):
            return a.to(self.dt)
        return _T(a).to(self.dt) if np.asarray(a).dtype != np.float64 else torch.from_numpy(np.ascontiguousarray(a)).to(self.dt)

    def w(self, name):
        return _T(self.arg[name]).to(self.dt)

    def bn(self, x, name, fix_gamma=False):
        g = torch.ones(x.shape[1], dtype=self.dt) if fix_gamma else self.w(name + '_gamma')
        return F.batch_norm(x, self.T(self.aux[name + '_moving_mean']), self.T(self.aux[name + '_moving_var']), g,
                            self.w(name + '_beta'), False, 0.0, EPS)

    def conv(self, x, name, k, stride=1, dilate=1, bias=True, pad=None):
        if k == 1:
            dilate = 1
        if pad is None:
            pad = ((k - 1) * dilate + 1) // 2          # sym_common.py:117-121
        return F.conv2d(x, self.w(name + '_weight'), self.w(name + '_bias') if bias else None, stride, pad, dilate)


def deformable_conv(p, x, name, num_filter, dilate, num_deformable_group=4):
    """sym_common.py:249-262: offset conv (with bias) then DeformableConvolution (no bias)."""
    off = p.conv(x, name + '_offset', 3, 1, dilate, bias=True)
    # the sampling operator receives fp32 maps in the reference (DeformableConvolution's im2col), the contraction is dense
    col = oracle.deform_im2col(x.numpy(), off.numpy(), 3, 3, dilate, 1, dilate, num_deformable_group)
    npdt = np.float64 if p.dt == torch.float64 else np.float32
    w = p.arg[name + '_weight'].reshape(num_filter, -1).astype(npdt)
    n, _, h, wd = off.shape
    out = np.stack([w @ col[i].astype(npdt) for i in range(n)], 0).reshape(n, num_filter, h, wd)
    return p.T(out)


def resnet_backbone(p, data, prefix='', need_part=False, add_dcn=True, stages=4):
    x = p.bn(data, prefix + 'bn_data', fix_gamma=True)
    x = p.conv(x, prefix + 'conv0', 7, 2, bias=False, pad=3)
    x = F.relu(p.bn(x, prefix + 'bn0'))
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    dilate = 1
    for s in range(1, stages + 1):
        nf = FILTERS[s - 1]
        for u in range(1, UNITS[s - 1] + 1):
            pre = '%sstage%d_unit%d_' % (prefix, s, u)
            stride = 2 if (u == 1 and s > 1) else 1
            inc_dilate = (u == 1 and s == 4)                  # inv_resolution=16, resnet.py:33-34
            unit_dilate = dilate
            if inc_dilate:
                stride = 1
                dilate = dilate * 2
            dcn = add_dcn and u >= UNITS[s - 1] - DEFORMABLE_UNITS[s - 1] + 1
            relu1 = F.relu(p.bn(x, pre + 'bn1'))
            conv1 = p.conv(relu1, pre + 'conv1', 1, 1, bias=False)
            relu2 = F.relu(p.bn(conv1, pre + 'bn2'))
            if dcn:
                conv2 = deformable_conv(p, relu2, pre + 'conv2', nf // 4, unit_dilate)
            else:
                conv2 = p.conv(relu2, pre + 'conv2', 3, stride, unit_dilate, bias=False)
            relu3 = F.relu(p.bn(conv2, pre + 'bn3'))
            conv3 = p.conv(relu3, pre + 'conv3', 1, 1, bias=False)
            shortcut = x if u > 1 else p.conv(relu1, pre + 'sc', 1, stride, bias=False)
            x = conv3 + shortcut
        outs.append(x)
    if need_part:
        return outs
    return F.relu(p.bn(x, prefix + 'bn1'))


def get_resnet_v1(p, data, cfg):
    out = resnet_backbone(p, data, add_dcn=cfg.network.add_dcn)
    return F.relu(F.conv2d(out, p.w('feat_conv_3x3_weight'), p.w('feat_conv_3x3_bias'), 1, 6, 6))


def get_flownet(p, img_cur, img_ref):
    def conv(x, name, k, stride=1, pad=1, act=True):
        y = F.conv2d(x, p.w(name + '_weight'), p.w(name + '_bias'), stride, pad)
        return F.leaky_relu(y, 0.1) if act else y

    def deconv_crop(x, name, like):
        y = F.conv_transpose2d(x, p.w(name + '_weight'), p.w(name + '_bias'), stride=2)
        return y[:, :, 1:1 + like.shape[2], 1:1 + like.shape[3]]

    data = torch.cat([p.T(img_cur) / 255.0, p.T(img_ref) / 255.0], 1)
    x = F.avg_pool2d(data, 2, 2, ceil_mode=True)
    r1 = conv(x, 'flow_conv1', 7, 2, 3)
    r2 = conv(r1, 'conv2', 5, 2, 2)
    r3 = conv(r2, 'conv3', 5, 2, 2)
    r4 = conv(r3, 'conv3_1', 3)
    r5 = conv(r4, 'conv4', 3, 2)
    r6 = conv(r5, 'conv4_1', 3)
    r7 = conv(r6, 'conv5', 3, 2)
    r8 = conv(r7, 'conv5_1', 3)
    r9 = conv(r8, 'conv6', 3, 2)
    r10 = conv(r9, 'conv6_1', 3)
    f6 = conv(r10, 'Convolution1', 3, act=False)
    c2 = torch.cat([r8, F.leaky_relu(deconv_crop(r10, 'deconv5', r8), 0.1), deconv_crop(f6, 'upsample_flow6to5', r8)], 1)
    f5 = conv(c2, 'Convolution2', 3, act=False)
    c3 = torch.cat([r6, F.leaky_relu(deconv_crop(c2, 'deconv4', r6), 0.1), deconv_crop(f5, 'upsample_flow5to4', r6)], 1)
    f4 = conv(c3, 'Convolution3', 3, act=False)
    c4 = torch.cat([r4, F.leaky_relu(deconv_crop(c3, 'deconv3', r4), 0.1), deconv_crop(f4, 'upsample_flow4to3', r4)], 1)
    f3 = conv(c4, 'Convolution4', 3, act=False)
    c5 = torch.cat([r2, F.leaky_relu(deconv_crop(c4, 'deconv2', r2), 0.1), deconv_crop(f3, 'upsample_flow3to2', r2)], 1)
    c5 = F.avg_pool2d(c5, 2, 2, ceil_mode=True)
    flow = conv(c5, 'Convolution5', 3, act=False) * 2.5
    scale = F.conv2d(c5, p.w('Convolution5_scale_weight'), p.w('Convolution5_scale_bias'))
    return flow, scale


def nq_logits(p, warp_feat, conv_feat):
    x = torch.cat([p.T(warp_feat), p.T(conv_feat)], 0)
    x = F.relu(F.conv2d(x, p.w('Nq_conv1_weight'), p.w('Nq_conv1_bias'), 1, 1))
    x = F.relu(F.conv2d(x, p.w('Nq_conv2_weight'), p.w('Nq_conv2_bias')))
    return F.conv2d(x, p.w('Nq_conv3_weight'), p.w('Nq_conv3_bias'))


def embed(p, conv_feat, warp_feat):
    """get_embednet on Concat(conv_feat, warp_feat) (symbols/resnet_v1_101_flownet_rfcn.py:118-135);
    row 0 embeds the current feature, row 1 the warped one."""
    x = torch.cat([p.T(conv_feat), p.T(warp_feat)], 0)
    x = F.relu(F.conv2d(x, p.w('em_conv1_weight'), p.w('em_conv1_bias')))
    x = F.relu(F.conv2d(x, p.w('em_conv2_weight'), p.w('em_conv2_bias'), 1, 1))
    return F.conv2d(x, p.w('em_conv3_weight'), p.w('em_conv3_bias'))


def small_net_feature(p, data_cur):
    img = F.avg_pool2d(p.T(data_cur), 4, 4, ceil_mode=True)
    feats = resnet_backbone(p, img, prefix='small_net_', need_part=True, add_dcn=False, stages=1)
    return F.conv2d(feats[0], p.w('fuse_reduce_add_weight'), p.w('fuse_reduce_add_bias'), 1, 1)


def head_maps(p, conv_feat, cfg):
    """RPN class probabilities + de-normalised deltas, and the two R-FCN score maps."""
    A = cfg.network.NUM_ANCHORS
    conv_feat = p.T(conv_feat)
    rpn_feat, rfcn_feat = conv_feat[:, :512], conv_feat[:, 512:]
    cls_score = F.conv2d(rpn_feat, p.w('rpn_cls_score_weight'), p.w('rpn_cls_score_bias'))
    bbox = F.conv2d(rpn_feat, p.w('rpn_bbox_pred_weight'), p.w('rpn_bbox_pred_bias'))
    if cfg.network.NORMALIZE_RPN:   # operator_py/rpn_inv_normalize.py:19-26
        std = torch.tensor(cfg.network.ANCHOR_STDS, dtype=torch.float32).to(p.dt).repeat(A).view(1, -1, 1, 1)
        mean = torch.tensor(cfg.network.ANCHOR_MEANS, dtype=torch.float32).to(p.dt).repeat(A).view(1, -1, 1, 1)
        bbox = bbox * std + mean
    n, _, h, w = cls_score.shape
    prob = torch.softmax(cls_score.reshape(n, 2, A * h, w), 1).reshape(n, 2 * A, h, w)
    cls_map = F.conv2d(rfcn_feat, p.w('rfcn_cls_weight'), p.w('rfcn_cls_bias'))
    box_map = F.conv2d(rfcn_feat, p.w('rfcn_bbox_weight'), p.w('rfcn_bbox_bias'))
    return prob, bbox, cls_map, box_map


def detect_from_maps(prob, bbox, cls_map, box_map, im_info, cfg, out=None):
    """Proposal + R-FCN head on fp32 maps (the operators' input dtype; a float64 graph's maps are rounded here).  With `out`, the
    anchor index behind every ROI row is recorded too (`roi_anchor`: index ((h * W) + w) * A + a of multi_proposal.cu:57-67)."""
    rois, _, order, keep, nkeep = oracle.proposal(prob.numpy(), bbox.numpy(), im_info, cfg.network.RPN_FEAT_STRIDE,
                                                  cfg.network.ANCHOR_SCALES, cfg.network.ANCHOR_RATIOS, cfg.TEST.RPN_PRE_NMS_TOP_N,
                                                  cfg.TEST.RPN_POST_NMS_TOP_N, cfg.TEST.RPN_NMS_THRESH, cfg.TEST.RPN_MIN_SIZE,
                                                  return_debug=True)
    if out is not None:
        out['roi_anchor'] = roi_anchor_index(order, keep, nkeep, rois.shape[0])
    cls_prob, _, bbox_pred = oracle.rfcn_head(cls_map.numpy(), box_map.numpy(), rois)
    return rois, cls_prob, bbox_pred


def roi_anchor_index(order, keep, nkeep, rows):
    """Which anchor each output row of Proposal is: row i of image b = sorted position keep[b, i % nkeep[b]] (the cyclic pad of
    multi_proposal.cu:374-386) = anchor order[b, that position]."""
    B = order.shape[0]
    post = rows // B
    out = np.empty(rows, np.int64)
    for b in range(B):
        k = keep[b, np.arange(post) % max(int(nkeep[b]), 1)]
        out[b * post:(b + 1) * post] = order[b, k]
    return out


def key_forward(cfg, arg, aux, data, data_key_old, feat_key_old, im_info, dtype=torch.float32):
    """get_key_test_symbol.  All inputs numpy; returns a dict of numpy stage outputs (in `dtype` for the dense stages)."""
    p = Params(arg, aux, dtype)
    with torch.no_grad():
        out = {}
        conv_feat = get_resnet_v1(p, p.T(data), cfg)
        out['backbone_feat'] = conv_feat.numpy()
        c, h, w = feat_key_old.shape[1:]
        is_first = (c == 1024 and h == 1 and w == 1)            # choose_old_key_feat.py:27
        if not is_first:
            flow, scale_map = get_flownet(p, p.T(data), p.T(data_key_old))
            out['flow'], out['scale_map'] = flow.numpy(), scale_map.numpy()
            warp = oracle.warp_bilinear(feat_key_old, out['flow'], mul=out['scale_map'])
            out['warp'] = warp
            if cfg.network.add_Nq_net:                          # :310-311
                logits = nq_logits(p, _T(warp), conv_feat)
                out['nq_logits'] = logits.numpy()
                conv_feat = p.T(oracle.aggregate_softmax2(warp, conv_feat.numpy(), out['nq_logits']))
            elif cfg.network.add_Fgfa_net:                      # :312-313, Fgfa_net :132-148
                out['embed'] = embed(p, conv_feat, _T(warp)).numpy()
                conv_feat = p.T(oracle.aggregate_cosine(warp, conv_feat.numpy(), out['embed'][1:2], out['embed'][0:1]))
            else:                                               # :314-315
                conv_feat = 0.5 * (p.T(warp) + conv_feat)
        out['choose_feat_output'] = conv_feat.numpy()
        prob, bbox, cls_map, box_map = head_maps(p, conv_feat, cfg)
        out.update(rpn_cls_prob=prob.numpy(), rpn_bbox_pred=bbox.numpy(), cls_map=cls_map.numpy(), box_map=box_map.numpy())
        rois, cls_prob, bbox_pred = detect_from_maps(prob, bbox, cls_map, box_map, im_info, cfg, out)
        out.update(rois_output=rois, cls_prob_reshape_output=cls_prob[None], bbox_pred_reshape_output=bbox_pred[None])
        return out


def cur_forward(cfg, arg, aux, data, feat_key, motion_vector, res_diff, im_info, dtype=torch.float32):
    p = Params(arg, aux, dtype)
    with torch.no_grad():
        out = {}
        small = small_net_feature(p, p.T(data))
        out['small_feat'] = small.numpy()
        conv_feat = oracle.warp_bilinear(feat_key, motion_vector, add=out['small_feat'], res=res_diff,
                                         res_w=arg['rnet_conv0_weight'], res_b=arg['rnet_conv0_bias'])
        out['conv_feat'] = conv_feat
        prob, bbox, cls_map, box_map = head_maps(p, _T(conv_feat), cfg)
        out.update(rpn_cls_prob=prob.numpy(), rpn_bbox_pred=bbox.numpy(), cls_map=cls_map.numpy(), box_map=box_map.numpy())
        rois, cls_prob, bbox_pred = detect_from_maps(prob, bbox, cls_map, box_map, im_info, cfg, out)
        out.update(rois_output=rois, cls_prob_reshape_output=cls_prob[None], bbox_pred_reshape_output=bbox_pred[None])
        return out


def batch_forward(cfg, arg, aux, data_key, data_other, im_info):
    """get_batch_test_symbol (:661-751): key frame + N others, tile_as, MultiProposal over the batch."""
    p = Params(arg, aux)
    with torch.no_grad():
        out = {}
        n = data_other.shape[0]
        feat_key = get_resnet_v1(p, _T(data_key), cfg)
        flow, scale_map = get_flownet(p, _T(data_other), _T(np.tile(data_key, (n, 1, 1, 1))))
        out['flow'], out['scale_map'], out['backbone_feat'] = flow.numpy(), scale_map.numpy(), feat_key.numpy()
        feat_other = oracle.warp_bilinear(np.tile(out['backbone_feat'], (n, 1, 1, 1)), out['flow'], mul=out['scale_map'])
        out['warp'] = feat_other
        conv_feat = torch.cat([feat_key, _T(feat_other)], 0)
        prob, bbox, cls_map, box_map = head_maps(p, conv_feat, cfg)
        out.update(rpn_cls_prob=prob.numpy(), rpn_bbox_pred=bbox.numpy(), cls_map=cls_map.numpy(), box_map=box_map.numpy())
        rois, cls_prob, bbox_pred = detect_from_maps(prob, bbox, cls_map, box_map, im_info, cfg)
        out.update(rois_output=rois, cls_prob_reshape_output=cls_prob[None], bbox_pred_reshape_output=bbox_pred[None])
        return out
