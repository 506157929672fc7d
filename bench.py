#!/usr/bin/env python
"""LSFA per-frame inference benchmark on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): frames/sec at 1000x600, key_interval = 10, synthetic VID-shaped clips,
random-init weights of the trained LSFA architecture (ResNet-101 + DCN + FlowNet + Nq + small net
+ R-FCN head), fp32.  Workload = BASELINE.json configs[1] ("dff_rfcn ResNet-101 LSFA, 1 clip
key_interval=10 on 1xMI355X, fp32").

A STEP is one key-frame interval of one clip: 1 key frame (ResNet-101 + FlowNet + flow warp x
scale map + Nq aggregation + heads + Proposal + PSROI + detection NMS) followed by
key_interval-1 non-key frames (small net + MV warp + residual + heads + Proposal + PSROI +
detection NMS).  Frames, motion vectors and residuals are resident in HBM before the timed
region; every frame's detections are copied to pinned host memory inside it.

N > 1: launched by torch.distributed.run, one rank per GPU; rank r runs clip r (clips are
independent: "scaling": "weak", no data-path collective); the only collective is the final
all_gather of per-frame detection counts over RCCL, outside the per-frame path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WARP_BYTES = lambda C, HW: (3 * C * HW + 2 * HW) * 4  # feat + (scale map | small-net feature) + out, + flow
HBM_PEAK_GBS = 8000.0                                   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HBM-side bytes per warp_kernel launch from rocprofv3 PMC passes (profiles/r1/warp_kernel_pmc_and_duration.txt,
# 1024x38x63): FETCH_SIZE 11,845 KiB x2 (the guide's gfx950 correction: 128-B requests are tallied as 64 B;
# uncalibrated for this kernel's 4/8-byte accesses) + WRITE_SIZE 9,699 KiB.  Only valid at that shape.
WARP_TRAFFIC_BYTES_38x63 = int((2 * 11844.9 + 9699.0) * 1024)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--interval', type=int, default=10)
    ap.add_argument('--height', type=int, default=600)
    ap.add_argument('--width', type=int, default=1000)
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='issue every launch eagerly instead of replaying hipGraphs')
    ap.add_argument('--no-prefetch', action='store_true', help='do not overlap the next frame\'s small net with this frame\'s tail')
    ap.add_argument('--lanes', type=int, default=2,
                    help='non-key frames of a segment alternate over this many streams while the next key frame runs on '
                         'its own stream (lsfa_amd.core.graphs.FramePipeline); 0 = strictly serial frames')
    ap.add_argument('--lookahead', action='store_true',
                    help='queue each key frame ahead of the non-key frames that precede it in display order')
    ap.add_argument('--no-flow-stream', action='store_true', help='FlowNet after the backbone on the key stream instead of beside it')
    ap.add_argument('--no-tuned-gemms', action='store_true',
                    help='library-default GEMM heuristics instead of lsfa_amd/tuned/gemm_gfx950.csv (lsfa_amd.tuning)')
    ap.add_argument('--cpu-budget-s', type=float, default=20.0)
    ap.add_argument('--max-unique-steps', type=int, default=16, help='distinct intervals of frames kept in HBM')
    return ap.parse_args()


class Runner(object):
    """One clip stream on one GPU: the pred_eval frame loop (dff_rfcn/core/tester.py:237-281)."""

    def __init__(self, args, rank, device):
        from lsfa_amd import hip
        from lsfa_amd.config.config import lsfa_test_config
        from lsfa_amd.symbols import params as P
        from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn
        from lsfa_amd.utils.synthetic import SyntheticClip
        self.hip, self.args, self.device = hip, args, device
        self.cfg = cfg = lsfa_test_config(key_frame_interval=args.interval)
        self.arg, self.aux = P.init_params(cfg, seed=0)
        dt = torch.float32 if args.dtype == 'f32' else torch.bfloat16
        net = resnet_v1_101_flownet_rfcn(cfg)
        self.key = net.get_key_test_symbol(cfg).bind(self.arg, self.aux, device, dt)
        self.cur = net.get_cur_test_symbol(cfg).bind(self.arg, self.aux, device, dt)
        self.K = args.interval
        self.nsteps_unique = min(args.max_unique_steps, args.steps + args.warmup)
        self.clip = SyntheticClip(rank, self.nsteps_unique * self.K + 1, args.height, args.width, self.K)
        self.im_info = torch.from_numpy(self.clip.im_info()).to(device)
        # frames resident in HBM: frame 0 primes the recurrence, then nsteps_unique intervals
        self.frames = [self.clip.frame(f, device) for f in range(self.nsteps_unique * self.K + 1)]
        self.mv, self.res = {}, {}
        for s in range(self.nsteps_unique):
            kf = 1 + s * self.K
            for i in range(1, self.K):
                self.mv[kf + i] = self.clip.motion_vector(kf + i, kf, device)
                self.res[kf + i] = self.clip.res_diff(kf + i, device)
        R, ncls = cfg.TEST.RPN_POST_NMS_TOP_N, cfg.dataset.NUM_CLASSES
        self.host_dets = torch.empty((self.K, ncls, R, 5), dtype=torch.float64).pin_memory()
        self.host_counts = torch.empty((self.K, ncls), dtype=torch.int32).pin_memory()
        from lsfa_amd.core.graphs import FrameGraphs, FramePipeline
        if args.lanes > 0:
            self.fg = FramePipeline(self.key, self.cur, cfg, args.height, args.width, device,
                                    use_graphs=not args.no_graph, lanes=args.lanes,
                                    flow_stream=not args.no_flow_stream, lookahead=args.lookahead)
        else:
            self.fg = FrameGraphs(self.key, self.cur, cfg, args.height, args.width, device, use_graphs=not args.no_graph,
                                  prefetch=not args.no_prefetch)

    @property
    def feat(self):
        return self.fg.feat

    def prime(self):
        """Frame 0 of the clip (flag 0: no aggregation) sets up feat_key / data_key; then the two
        per-frame launch sequences are captured into hipGraphs (untimed)."""
        self.fg.first_frame(self.frames[0])
        self.fg.capture()

    def _deliver(self, bufs, slot):
        self.host_dets[slot].copy_(bufs[0], non_blocking=True)
        self.host_counts[slot].copy_(bufs[1], non_blocking=True)

    def step(self, s, fg=None):
        """One key-frame interval: key frame (flag 1) + K-1 non-key frames (flag 2)."""
        fg = fg or self.fg
        kf = 1 + (s % self.nsteps_unique) * self.K
        if hasattr(fg, 'lanes'):      # pipelined: frames are queued in order, copies ride on each frame's stream
            fg.key_frame(self.frames[kf], deliver=lambda b: self._deliver(b, 0))
            for i in range(1, self.K):
                fg.cur_frame(self.frames[kf + i], self.mv[kf + i], self.res[kf + i],
                             deliver=lambda b, i=i: self._deliver(b, i))
            return
        nxt = lambda i: self.frames[kf + i + 1] if i + 1 < self.K else None   # the frame after a non-key frame, if non-key too
        self._deliver(fg.key_frame(self.frames[kf], nxt(0)), 0)
        for i in range(1, self.K):
            self._deliver(fg.cur_frame(self.frames[kf + i], self.mv[kf + i], self.res[kf + i], nxt(i)), i)

    def eager_profile_step(self, s):
        """The same interval issued eagerly with the per-kernel event hooks on (a captured graph has no
        per-launch events): the roofline leg."""
        from lsfa_amd.core.graphs import FrameGraphs
        eg = FrameGraphs(self.key, self.cur, self.cfg, self.args.height, self.args.width, self.device, use_graphs=False,
                         prefetch=False)
        eg.feat_old.copy_(self.fg.feat_old)
        eg.data_key_old.copy_(self.fg.data_key_old)
        eg.feat = self.fg.feat.clone()
        self.step(s, eg)     # warm
        torch.cuda.synchronize()
        self.hip.prof_enable(True, ops=['warp_bilinear', 'aggregate', 'rfcn_head', 'proposal', 'det_postprocess'])
        self.hip.prof_read()
        self.step(s, eg)
        torch.cuda.synchronize()
        prof = self.hip.prof_read()
        self.hip.prof_enable(False)
        return prof

    def cpu_baseline(self, budget_s):
        """The oracle's statement of the same step (torch-CPU convs + C kernels), timed on the host."""
        import oracle
        from oracle import graph_ref
        cfg, K = self.cfg, self.K
        f = [self.frames[i].cpu().numpy() for i in range(0, K + 1)]
        im_info = self.clip.im_info()
        feat0 = self.feat.cpu().numpy()
        t0 = time.time()
        frames_done = 0
        out = graph_ref.key_forward(cfg, self.arg, self.aux, f[1], f[0], feat0, im_info)
        oracle.det_postprocess(out['rois_output'], out['bbox_pred_reshape_output'][0], out['cls_prob_reshape_output'][0],
                               self.args.height, self.args.width, 1.0)
        feat = out['choose_feat_output']
        frames_done += 1
        key_s = time.time() - t0
        nonkey_s = 0.0
        for i in range(1, K):
            t1 = time.time()
            o = graph_ref.cur_forward(cfg, self.arg, self.aux, f[1 + i], feat, self.mv[1 + i].cpu().numpy(),
                                      self.res[1 + i].cpu().numpy(), im_info)
            oracle.det_postprocess(o['rois_output'], o['bbox_pred_reshape_output'][0], o['cls_prob_reshape_output'][0],
                                   self.args.height, self.args.width, 1.0)
            nonkey_s += time.time() - t1
            frames_done += 1
            if time.time() - t0 > budget_s:
                break
        n_nonkey = frames_done - 1
        # per-interval time = 1 key + (K-1) non-key at the measured per-frame costs
        per_step = key_s + (nonkey_s / max(n_nonkey, 1)) * (K - 1)
        return {"value": round(K / per_step, 3), "unit": "frames/s", "cores": int(torch.get_num_threads()), "kind": "port",
                "sample": "1 key frame + %d non-key frame(s) of the same clip at %dx%d through oracle/graph_ref.py "
                          "(torch-CPU fp32 convs, C oracle for warp/aggregate/proposal/psroi/nms); %.1f s key, %.2f s per "
                          "non-key frame, extrapolated to one %d-frame interval" % (n_nonkey, self.args.width,
                                                                                   self.args.height, key_s,
                                                                                   nonkey_s / max(n_nonkey, 1), K)}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    distributed = world > 1
    # test hook (one-GPU boxes): LSFA_BENCH_BACKEND=gloo LSFA_BENCH_ONE_DEVICE=1 runs every rank on cuda:0 and
    # the collectives on CPU tensors, to exercise the N>1 control flow without N GPUs
    backend = os.environ.get('LSFA_BENCH_BACKEND', 'nccl')
    if os.environ.get('LSFA_BENCH_ONE_DEVICE') == '1':
        local_rank = 0
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(local_rank)
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    coll_dev = ('cuda:%d' % local_rank) if backend == 'nccl' else 'cpu'
    device = 'cuda:%d' % local_rank
    torch.cuda.set_device(local_rank)
    torch.backends.cudnn.benchmark = os.environ.get('LSFA_MIOPEN_FIND', '1') == '1'

    from lsfa_amd import hip, tuning
    tuned = None if args.no_tuned_gemms else tuning.enable(tune_missing=True)
    r = Runner(args, rank, device)
    r.prime()
    def drain():
        if hasattr(r.fg, 'flush'):
            r.fg.flush()             # the pipeline queues a segment when the next key frame arrives: queue the last one
        torch.cuda.synchronize()     # device-wide: drains every stream of the frame pipeline

    for s in range(args.warmup):
        r.step(s)
    drain()

    def barrier():
        if distributed:
            dist.barrier(device_ids=[local_rank]) if backend == 'nccl' else dist.barrier()

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        r.step(s)
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = r.eager_profile_step(args.warmup) if rank == 0 else None

    if distributed:
        tt = torch.tensor([elapsed], device=coll_dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # the final gather of detections (here: last interval's per-frame counts) over RCCL
        counts = r.host_counts.to(coll_dev)
        gathered = [torch.empty_like(counts) for _ in range(world)]
        dist.all_gather(gathered, counts)
        total_dets = int(sum(int(g.sum().item()) for g in gathered))
    else:
        total_dets = int(r.host_counts.sum().item())

    if rank == 0:
        K = args.interval
        frames = world * args.steps * K
        fh, fw = int(np.ceil(args.height / 16.0)), int(np.ceil(args.width / 16.0))
        warp_ms, warp_n = prof['warp_bilinear']
        bytes_per_launch = WARP_BYTES(1024, fh * fw)
        achieved = bytes_per_launch * warp_n / (warp_ms * 1e-3) / 1e9 if warp_ms > 0 else 0.0
        line = {
            "metric": "frames/sec/GPU at 1000x600 key_interval=10 (whole-job frames/s)",
            "value": round(frames / elapsed, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "dff_rfcn ResNet-101 LSFA (DCN + FlowNet + Nq + small net + R-FCN), 1 clip per GPU, "
                                   "key_interval=%d, %dx%d, %s; step = 1 key + %d non-key frames" %
                                   (K, args.width, args.height, args.dtype, K - 1),
                       "frames_per_step": K, "ms_per_frame": round(elapsed / (args.steps * K) * 1e3, 3),
                       "parallelism": "clip-parallel x%d" % world, "detections_last_interval": total_dets,
                       "launch": "eager" if args.no_graph else "hipGraph replay per frame",
                       "gemm_solutions": "library default" if tuned is None else
                                         ("lsfa_amd/tuned/gemm_gfx950.csv" if tuned else "tuned on first use (shipped file rejected)"),
                       "pipeline": ("key stream%s + %d non-key lanes%s" % (
                           "" if args.no_flow_stream else " + FlowNet/tail stream", args.lanes,
                           ", next key frame queued ahead of the segment before it" if args.lookahead else ""))
                       if args.lanes > 0 else "serial"},
            "roofline": {"bound": "hbm", "kernel": "warp_kernel (lsfa_warp_bilinear: MV/flow warp + fused epilogue)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": WARP_TRAFFIC_BYTES_38x63 if (fh, fw) == (38, 63) else None,
                         "launches": warp_n, "avg_us": round(warp_ms * 1e3 / max(warp_n, 1), 2),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "measured": "HIP events around each launch, same frames re-issued eagerly right after the "
                                     "timed region (graph replays carry no per-launch events)",
                         "other_ops_avg_us": {k: round(v[0] * 1e3 / max(v[1], 1), 2) for k, v in prof.items() if v[1]}},
        }
        if (args.height, args.width) == (600, 1000):
            # algorithmic FLOPs of the dense contractions, key frame 392.0 G, non-key frame 17.3 G
            gflop = 392.0 + 17.3 * (K - 1)
            peak = 157.0 if args.dtype == 'f32' else 2500.0
            tf = gflop * world / (elapsed / args.steps) / 1e3
            line["roofline_dense"] = {"bound": "mfma", "scope": "all dense contractions of one interval (library MFMA kernels), "
                                      "algorithmic FLOPs / step time, per GPU", "achieved": round(tf / world, 1), "peak": peak,
                                      "unit": "TFLOP/s", "frac": round(tf / world / peak, 4), "gflop_per_step": round(gflop, 1)}
        if not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = r.cpu_baseline(args.cpu_budget_s)
            except Exception as e:  # the baseline is a reported extra; never lose the bench line to it
                line["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": int(torch.get_num_threads()),
                                        "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(line))
    if distributed:
        dist.barrier(device_ids=[local_rank]) if backend == 'nccl' else dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
