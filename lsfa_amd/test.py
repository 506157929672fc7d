#!/usr/bin/env python
"""Python-3 counterpart of dff_rfcn/test.py (:24-58) + experiments/dff_rfcn/dff_rfcn_test.py:
parse --cfg, build the test symbols, shard videos over the ranks, run pred_eval, gather, report.

    python -m lsfa_amd.test [--cfg YAML] [--clips N] [--frames F] [--height H] [--width W] [--prefix P --epoch E]
    python -m torch.distributed.run --nproc-per-node 8 -m lsfa_amd.test --clips 8

Without a dataset or weights in the image, clips are synthetic (lsfa_amd.utils.synthetic) and the
weights are the seeded random init unless --prefix points at an MXNet `.params` checkpoint.
"""
import argparse
import logging
import os
import time

import torch
import torch.distributed as dist

from lsfa_amd.config.config import config, lsfa_test_config, update_config, update_network_config
from lsfa_amd.function.test_rcnn import test_rcnn
from lsfa_amd.symbols import params as P
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
from lsfa_amd.utils.load_model import load_param
from lsfa_amd.utils.synthetic import synthetic_roidb


def parse_args():
    ap = argparse.ArgumentParser(description='Test a LSFA (dff_rfcn) network')
    ap.add_argument('--cfg', help='experiment configure file name', default=None, type=str)
    ap.add_argument('--thresh', help='valid detection threshold', default=1e-4, type=float)
    ap.add_argument('--clips', type=int, default=2)
    ap.add_argument('--frames', type=int, default=24)
    ap.add_argument('--height', type=int, default=600)
    ap.add_argument('--width', type=int, default=1000)
    ap.add_argument('--interval', type=int, default=None, help='override TEST.KEY_FRAME_INTERVAL')
    ap.add_argument('--prefix', default=None, help='MXNet checkpoint prefix (prefix-%%04d.params)')
    ap.add_argument('--epoch', type=int, default=0)
    ap.add_argument('--serial', action='store_true', help='reference-shaped serial frame loop (pred_eval) instead of the '
                                                          'stream-pipelined one')
    ap.add_argument('--segment', type=int, default=0,
                    help='non-key frames per batched pass of the pipelined loop (-1: a whole segment, KEY_FRAME_INTERVAL - 1; 0: frame by '
                         'frame, every detection equal to the serial loop\'s bit for bit)')
    ap.add_argument('--key-group', type=int, default=1, help='key frames whose backbone + FlowNet run in one pass (look-ahead through the loader)')
    ap.add_argument('--out', default=None, help='rank 0 saves the gathered detection rows (n,7) here (.npy)')
    ap.add_argument('--shards-out', default=None, help='rank 0 saves which videos each rank ran (a JSON list per rank, gathered from the ranks themselves)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help='f32: every fp32 product from two fp16 pieces (fp32 accuracy); bf16: one bf16 product per fp32 product (BASELINE configs[2])')
    return ap.parse_args()


def main():
    args = parse_args()
    if args.cfg:
        cfg = update_config(args.cfg, config)
        update_network_config(cfg)
    else:
        cfg = lsfa_test_config()
    if args.interval:
        cfg.TEST.KEY_FRAME_INTERVAL = args.interval
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # one-GPU boxes: LSFA_BENCH_BACKEND=gloo LSFA_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and runs the final
    # gather over gloo, so the N > 1 control flow (sharding, gather, merge) can be exercised without N GPUs
    backend = os.environ.get('LSFA_BENCH_BACKEND', 'nccl')
    if os.environ.get('LSFA_BENCH_ONE_DEVICE') == '1':
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # LSFA_BENCH_FORCE_DIST=1: initialise the process group at world size 1 too (a one-GPU box then runs RCCL's init and the final
    # gather's collectives on device tensors: tests/test_multirank_gpu.py)
    if world > 1 or (os.environ.get('LSFA_BENCH_FORCE_DIST') == '1' and 'RANK' in os.environ):
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    logging.basicConfig(level=logging.INFO, format='%(asctime)s %(message)s')
    logger = logging.getLogger('lsfa')
    roidb = synthetic_roidb(args.clips, args.frames, args.height, args.width, cfg.TEST.KEY_FRAME_INTERVAL)
    if args.prefix:
        arg_params, aux_params = load_param(args.prefix, args.epoch, process=True)
        net = resnet_v1_101_flownet_rfcn(cfg)
        for getter in (net.get_key_test_symbol, net.get_cur_test_symbol):
            getter(cfg)
            net.init_weight(cfg, arg_params, aux_params)
    else:
        arg_params, aux_params = P.init_params(cfg, seed=0)
    t0 = time.time()
    rows, frame_ids = test_rcnn(cfg, roidb, arg_params, aux_params, device='cuda:%d' % local_rank, thresh=args.thresh,
                                logger=logger, pipeline=not args.serial,
                                segment=(cfg.TEST.KEY_FRAME_INTERVAL - 1 if args.segment < 0 else args.segment), key_group=args.key_group,
                                dtype=torch.float32 if args.dtype == 'f32' else torch.bfloat16)
    torch.cuda.synchronize()
    dt = time.time() - t0
    if args.shards_out:
        # what every rank actually ran (the frame ids its own loader handed out -> video indices), gathered from the ranks - not recomputed
        mine = sorted(set(int(f) // args.frames for f in frame_ids))
        seen = [None] * world
        if dist.is_initialized():
            dist.all_gather_object(seen, mine)
        else:
            seen = [mine]
        if not dist.is_initialized() or dist.get_rank() == 0:
            import json
            with open(args.shards_out, 'w') as f:
                json.dump(seen, f)
    if not dist.is_initialized() or dist.get_rank() == 0:
        total = args.clips * args.frames
        print('%d clips x %d frames on %d GPU(s): %d detections, %.1f frames/s incl. setup' % (
            args.clips, args.frames, world, len(rows), total / dt))
        if args.out:
            import numpy as np
            np.save(args.out, rows)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
