"""Hot-path configuration, mirroring dff_rfcn/config/config.py (keys consumed at test time).

`config` is the module-level attribute dictionary the reference calls `config`;
`update_config(yaml)` applies a yaml the same way (unknown top-level keys raise,
config.py:188-209) and `update_network_config()` derives nettype / num_layer /
PIXEL_MEANS from `network.pretrained` (config.py:170-186).  Training-only keys of
the reference's yaml files are accepted and stored but nothing here reads them.
"""
import copy

import numpy as np
import yaml


class AttrDict(dict):
    """EasyDict stand-in (easydict is not installed): attribute access on a dict, recursively."""

    def __init__(self, d=None, **kw):
        super(AttrDict, self).__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            v = AttrDict(v)
        super(AttrDict, self).__setitem__(k, v)

    __setattr__ = __setitem__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __deepcopy__(self, memo):
        return AttrDict({k: copy.deepcopy(v, memo) for k, v in self.items()})


def default_config():
    c = AttrDict()
    c.MXNET_VERSION = ''
    c.output_path = ''
    c.symbol = 'resnet_v1_101_flownet_rfcn'
    c.gpus = '0'
    c.CLASS_AGNOSTIC = True
    c.SCALES = [(600, 1000)]
    c.default = AttrDict(frequent=20, kvstore='device')
    n = AttrDict()
    n.pretrained = ''
    n.pretrained_flow = ''
    n.nettype = ''
    n.num_layer = None
    n.pretrained_epoch = 0
    n.PIXEL_MEANS = np.array([0, 0, 0])
    n.PIXEL_SCALE = 1
    n.IMAGE_STRIDE = 0
    n.RPN_FEAT_STRIDE = 16
    n.RCNN_FEAT_STRIDE = 16
    n.FIXED_PARAMS = ['gamma', 'beta']
    n.ANCHOR_SCALES = (8, 16, 32)
    n.ANCHOR_RATIOS = (0.5, 1, 2)
    n.NORMALIZE_RPN = True
    n.ANCHOR_MEANS = (0.0, 0.0, 0.0, 0.0)
    n.ANCHOR_STDS = (0.1, 0.1, 0.4, 0.4)
    n.NUM_ANCHORS = 9
    n.DFF_FEAT_DIM = 1024
    n.rnet_num_conv = None
    n.fnet_type = None
    n.fuse_type = 'add'
    n.res_diff_bn = False
    n.add_dcn = False
    n.add_small_net = False
    n.small_net_bn_before_fuse = False
    n.small_net_scale_before_fuse = False
    n.small_net_stride = 4
    n.small_net_fuse_type = 'add'
    n.add_Nq_net = False
    n.add_Fgfa_net = False
    c.network = n
    c.dataset = AttrDict(dataset='ImageNetVID', image_set='DET_train_30classes+VID_train_15frames',
                         test_image_set='VID_val_videos', root_path='./data', dataset_path='./data/ILSVRC2015',
                         NUM_CLASSES=31)
    c.TRAIN = AttrDict()
    t = AttrDict()
    t.HAS_RPN = False
    t.BATCH_IMAGES = 1
    t.CXX_PROPOSAL = True
    t.RPN_NMS_THRESH = 0.7
    t.RPN_PRE_NMS_TOP_N = 6000
    t.RPN_POST_NMS_TOP_N = 300
    t.RPN_MIN_SIZE = n.RPN_FEAT_STRIDE
    t.NMS = 0.3
    t.KEY_FRAME_INTERVAL = 12
    t.max_per_image = 300
    t.test_epoch = 0
    c.TEST = t
    return c


config = default_config()


def update_network_config(cfg=None):
    cfg = config if cfg is None else cfg
    if 'resnet' in cfg.network.pretrained:
        cfg.network.PIXEL_MEANS = [0, 0, 0]
        cfg.network.PIXEL_SCALE = 1.0
        nettype, num_layer = cfg.network.pretrained.split('/')[-1].split('-')
        cfg.network.num_layer = int(float(num_layer))
        cfg.network.nettype = 'resnet'
    else:
        raise RuntimeError("unknow nettype")   # mobilenet backbones are out of scope


def update_config(config_file, cfg=None):
    cfg = config if cfg is None else cfg
    with open(config_file) as f:
        exp_config = yaml.safe_load(f)
    for k, v in exp_config.items():
        if k not in cfg:
            raise ValueError("key must exist in config.py")
        if isinstance(v, dict):
            if k == 'network' and 'PIXEL_MEANS' in v:
                v['PIXEL_MEANS'] = np.array(v['PIXEL_MEANS'])
            for vk, vv in v.items():
                cfg[k][vk] = vv
        elif k == 'SCALES':
            cfg[k][0] = tuple(v)
        else:
            cfg[k] = v
    return cfg


def lsfa_test_config(key_frame_interval=10):
    """The trained LSFA configuration (experiments/dff_rfcn/cfgs/
    resnet_v1_101_flownet_imagenet_vid_rfcn_end2end_ohem.yaml:49-60, TEST :143-156) as a fresh
    config object; KEY_FRAME_INTERVAL is BASELINE.json's 10 unless overridden."""
    import os
    c = default_config()
    here = os.path.dirname(os.path.abspath(__file__))
    update_config(os.path.join(here, 'resnet_v1_101_flownet_imagenet_vid_rfcn_end2end_ohem.yaml'), c)
    update_network_config(c)
    c.TEST.KEY_FRAME_INTERVAL = key_frame_interval
    return c
