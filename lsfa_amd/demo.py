#!/usr/bin/env python
"""Counterpart of dff_rfcn/demo.py (:63-158): run the key / non-key frame loop over one clip, print
the running mean time per frame like the reference's tic/toc loop, and report the detections that
score above 0.7 after per-class NMS.

    python -m lsfa_amd.demo                         # synthetic 1000x600 clip, random-init weights
    python -m lsfa_amd.demo --frames DIR [--mv DIR] [--prefix P --epoch E] [--out dets.json]

--frames: a directory of *.JPEG / *.jpg / *.png frames in display order (decoded with PIL; the
reference uses cv2.imread, :75).  --mv: one `<frame stem>.npz` per non-key frame holding `mv`
(H, W, 2) and `res` (H, W, 3) in source-image pixels, the arrays lib/utils/image.py:get_image reads
from the compressed stream; without it non-key frames propagate the key feature unchanged (zero
motion, zero residual), which is what the reference's own demo amounts to (it has no MV input).
Drawing boxes into images (draw_boxes, :150-156) is left to the caller: the output is JSON.
"""
import argparse
import glob
import json
import os
import time

import numpy as np
import torch

from lsfa_amd.config.config import config, lsfa_test_config, update_config, update_network_config
from lsfa_amd.core.graphs import FrameGraphs
from lsfa_amd.symbols import params as P
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
from lsfa_amd.utils.image import resize, transform, transform_mv_res
from lsfa_amd.utils.load_model import load_param
from lsfa_amd.utils.synthetic import SyntheticClip


class FrameDirClip(object):
    """Frames of one clip from a directory, preprocessed like the reference's demo (:73-82)."""

    def __init__(self, frame_dir, mv_dir, cfg):
        from PIL import Image
        names = sorted(sum((glob.glob(os.path.join(frame_dir, e)) for e in ('*.JPEG', '*.jpg', '*.jpeg', '*.png')), []))
        if not names:
            raise FileNotFoundError('no frames under %s' % frame_dir)
        self.names, self.mv_dir, self.cfg = names, mv_dir, cfg
        self._open = Image.open
        self.num_frames = len(names)
        f0, self.im_scale = self._load(0)
        self.height, self.width = f0.shape[2], f0.shape[3]

    def _load(self, i):
        cfg = self.cfg
        rgb = np.asarray(self._open(self.names[i]).convert('RGB'), dtype=np.float32)
        bgr = torch.from_numpy(np.ascontiguousarray(rgb[:, :, ::-1]))
        im, im_scale = resize(bgr, cfg.SCALES[0][0], cfg.SCALES[0][1], stride=cfg.network.IMAGE_STRIDE)
        return transform(im, cfg.network.PIXEL_MEANS, cfg.network.PIXEL_SCALE), im_scale

    def frame(self, i):
        return self._load(i)[0]

    def mv_res(self, i, key_i):
        fh, fw = -(-self.height // 16), -(-self.width // 16)
        if self.mv_dir is None:
            return torch.zeros(1, 2, fh, fw), torch.zeros(1, 3, fh, fw)
        stem = os.path.splitext(os.path.basename(self.names[i]))[0]
        z = np.load(os.path.join(self.mv_dir, stem + '.npz'))
        cfg = self.cfg
        return transform_mv_res(z['mv'], z['res'], self.im_scale, cfg.network.PIXEL_MEANS, cfg.network.PIXEL_SCALE)


class _Synthetic(object):
    def __init__(self, n, h, w):
        self.c = SyntheticClip(0, n, h, w)
        self.num_frames, self.height, self.width, self.im_scale = n, h, w, 1.0
        self.names = ['synthetic/%06d' % i for i in range(n)]

    def frame(self, i):
        return self.c.frame(i)

    def mv_res(self, i, key_i):
        return self.c.motion_vector(i, key_i), self.c.res_diff(i)


def parse_args():
    ap = argparse.ArgumentParser(description='LSFA demo: key / non-key frame loop over one clip')
    ap.add_argument('--cfg', default=None)
    ap.add_argument('--frames', default=None, help='directory of frames (default: a synthetic clip)')
    ap.add_argument('--mv', default=None, help='directory of per-frame .npz with mv / res arrays')
    ap.add_argument('--num', type=int, default=30, help='frames of the synthetic clip')
    ap.add_argument('--interval', type=int, default=10, help='key frame interval (demo.py:68)')
    ap.add_argument('--prefix', default=None)
    ap.add_argument('--epoch', type=int, default=0)
    ap.add_argument('--score', type=float, default=0.7, help='report threshold (demo.py:147)')
    ap.add_argument('--out', default=None, help='write the detections as JSON here')
    ap.add_argument('--no-graph', action='store_true')
    return ap.parse_args()


def main():
    args = parse_args()
    if args.cfg:
        cfg = update_config(args.cfg, config)
        update_network_config(cfg)
    else:
        cfg = lsfa_test_config()
    cfg.TEST.KEY_FRAME_INTERVAL = args.interval
    dev = 'cuda:0'
    clip = FrameDirClip(args.frames, args.mv, cfg) if args.frames else _Synthetic(args.num, 600, 1000)
    if args.prefix:
        arg_params, aux_params = load_param(args.prefix, args.epoch, process=True)
    else:
        arg_params, aux_params = P.init_params(cfg, seed=0)
    net = resnet_v1_101_flownet_rfcn(cfg)
    key = net.get_key_test_symbol(cfg).bind(arg_params, aux_params, dev)
    cur = net.get_cur_test_symbol(cfg).bind(arg_params, aux_params, dev)
    fg = FrameGraphs(key, cur, cfg, clip.height, clip.width, dev, thresh=args.score, use_graphs=not args.no_graph,
                     prefetch=False)
    fg.scale = float(clip.im_scale)
    fg.im_info[0, 2] = fg.scale
    classes = None       # class names live in the dataset (imdb.classes); ids are reported without one

    results, total, count = [], 0.0, 0
    for idx in range(clip.num_frames):
        data = clip.frame(idx).to(dev)
        mv, res = (None, None) if idx % args.interval == 0 else [t.to(dev) for t in clip.mv_res(idx, idx - idx % args.interval)]
        torch.cuda.synchronize()
        t0 = time.time()
        if idx == 0:
            dets, counts, _ = fg.first_frame(data)
            torch.cuda.synchronize()
            dets_h, counts_h = dets.cpu().numpy(), counts.cpu().numpy()
            fg.capture()                                   # the reference's "warm up" (:104-116)
            print('warmup done')
        else:
            dets, counts, _ = fg.key_frame(data) if idx % args.interval == 0 else fg.cur_frame(data, mv, res)
            dets_h, counts_h = dets.cpu().numpy(), counts.cpu().numpy()    # .cpu() is the per-frame sync
            total += time.time() - t0
            count += 1
            print('testing {} {:.4f}s'.format(clip.names[idx], total / count))
        frame_dets = []
        for j in range(1, dets_h.shape[0]):
            for x1, y1, x2, y2, s in dets_h[j, :counts_h[j]]:
                frame_dets.append({'class': classes[j] if classes else j, 'score': float(s),
                                   'box': [float(x1), float(y1), float(x2), float(y2)]})
        results.append({'frame': clip.names[idx], 'key': idx % args.interval == 0, 'dets': frame_dets})
    print('done: {} frames, {} detections above {:.2f}'.format(len(results), sum(len(r['dets']) for r in results),
                                                                args.score))
    if args.out:
        with open(args.out, 'w') as f:
            json.dump(results, f)
    return results


if __name__ == '__main__':
    main()
