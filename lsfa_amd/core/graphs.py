"""hipGraph capture of the two per-frame launch sequences (key frame, non-key frame).

A non-key frame is ~50 kernels of 2-150 us each and a key frame ~480: issued eagerly from
Python the host cannot keep the GPU fed (the reference has the same problem one level up: a
blocking .asnumpy() and 30 numpy NMS calls per frame, tester.py:138-152, :265-281).  Every entry
point of liblsfa_hip.so is allocation- and sync-free by contract (include/lsfa_hip.h), so a whole
frame — network forward + lsfa_det_postprocess — is captured once into a hipGraph and replayed.

State carried across frames stays on the device: the key graph's output feature is the cur
graph's `feat_key` input by pointer (no copy); the only per-frame host work is queuing the
copies of that frame's inputs into the graph's static buffers, one replay, and the asynchronous
copy of the detections to pinned host memory.

Software pipelining inside the graphs (`prefetch=True`): the small-net branch of a non-key frame
needs only that frame's image, while the tail of a frame (Proposal's single-workgroup
select/sort/NMS, the R-FCN head, the detection NMS) keeps one or a few CUs busy.  Each captured
graph therefore forks a second stream that computes the small-net feature of the NEXT frame while
the current frame's tail runs, and the next replay consumes it.
"""
import os
import sys

import torch

from lsfa_amd import hip
from lsfa_amd.core import streams


def _alloc_post(batch, ncls, R, device):
    """Detection output buffers of one frame slot: (B, ncls, R, 5) f64, (B, ncls) i32, (B, ncls, R) i32.
    Returns (all, public) where `public` drops the batch axis when B == 1 (the shapes callers had before
    several clips could advance together).  dets and counts live back to back in ONE allocation, reachable as
    `dets.lsfa_flat` (float64 view of both): a consumer takes a frame's results off the device with one copy."""
    n_d = batch * ncls * R * 5
    n_c = (batch * ncls * 4 + 7) // 8                       # the counts, in float64 slots
    flat = torch.zeros(n_d + n_c, dtype=torch.float64, device=device)
    dets = flat[:n_d].view(batch, ncls, R, 5)
    counts = flat[n_d:].view(torch.int32)[:batch * ncls].view(batch, ncls)
    full = (dets, counts, torch.full((batch, ncls, R), -1, dtype=torch.int32, device=device))
    public = tuple(t[0] for t in full) if batch == 1 else full
    full[0].lsfa_flat = flat
    public[0].lsfa_flat = flat
    return full, public


# r6: batched passes read their frames where the caller left them (hip.ImageTable: a device table of per-image pointers, rewritten by one tiny
# launch per pass) instead of from a static batch buffer filled by a staging copy (2 x 65 MB per nine-frame segment, 2 x 86 MB per pass of six key
# fronts: 2.4 % of frames/s, profiles/r6/tail_ablation.txt).  LSFA_STAGE_FRAMES=1 brings the copies back (A/B, and the fall-back for frames the
# table cannot take: not float32, not contiguous, another device).
ZERO_COPY_FRAMES = os.environ.get('LSFA_STAGE_FRAMES') != '1'


def _table_ok(t, shape, device):
    return t.is_cuda and t.device == device and t.dtype == torch.float32 and t.is_contiguous() and tuple(t.shape[-3:]) == tuple(shape)


def _stage_inputs(jobs):
    """Frame inputs into the static buffers a captured graph reads: one launch (lsfa_copy_many) when the tensors allow
    it (same device, fp32, contiguous, equal shapes), else plain copies."""
    ok = all(s.is_cuda and s.device == d.device and s.dtype == d.dtype and d.element_size() == 4 and s.shape == d.shape and
             s.is_contiguous() and d.is_contiguous() for d, s in jobs)
    if ok:
        for i in range(0, len(jobs), 32):
            hip.copy_many(jobs[i:i + 32])
    else:
        for d, s in jobs:
            d.copy_(s)


def _post_all(out, full, cfg, h, w, scale, thresh):
    """lsfa_det_postprocess_batch: every image of the batch in one launch pair (rois of image b are rows [b*R, (b+1)*R), MultiProposal's layout)."""
    B = full[0].shape[0]
    R = out['rois_output'].shape[0] // B
    bbox, cls = out['bbox_pred_reshape_output'].reshape(B * R, -1), out['cls_prob_reshape_output'].reshape(B * R, -1)
    hip.det_postprocess_batch(out['rois_output'], bbox, cls, B, h, w, scale, full, score_thresh=thresh, nms_thresh=cfg.TEST.NMS,
                              max_per_image=cfg.TEST.max_per_image, class_agnostic=cfg.CLASS_AGNOSTIC)


class FrameGraphs(object):
    def __init__(self, key_exec, cur_exec, cfg, height, width, device, thresh=1e-4, use_graphs=True, prefetch=True,
                 feat_shared=None, taps=False, batch=1, frames_by_table=False):
        """feat_shared: a caller-owned (1, DFF_FEAT_DIM, h, w) buffer the non-key graph reads the key feature
        from; such an instance is a non-key "lane" of a FramePipeline and never runs key frames.
        taps=True (parity tests): the stage outputs of the last key / non-key frame stay readable in
        `key_taps` / `cur_taps` (+ the network outputs in `key_out` / `cur_out`); under hipGraph replay they
        are the graphs' static buffers, so read or clone them on the frame's stream before the next replay."""
        self.key, self.cur, self.cfg = key_exec, cur_exec, cfg
        self.feat_shared = feat_shared
        # r6: a lane whose non-key frames run on channels-last maps takes the key feature channels-last (feat_shared_cl: the pipeline's hand-over
        # IS the transposing copy - set_key_feature - instead of a plain copy followed by a transposition in every pass)
        self.feat_shared_cl = None
        if feat_shared is not None and not prefetch and cur_exec is not None and cur_exec.cur_channels_last(cfg.network.DFF_FEAT_DIM):
            self.feat_shared_cl = torch.zeros((feat_shared.shape[0], feat_shared.shape[2], feat_shared.shape[3], feat_shared.shape[1]),
                                              device=feat_shared.device, dtype=torch.float32)
        self.want_taps = taps
        self.key_taps = self.cur_taps = self.key_out = self.cur_out = None
        self._small_valid = False      # small_cur holds the small-net feature of the frame cur_frame is about to get
        self.device = torch.device(device)
        self.use_graphs = use_graphs
        self.h, self.w = height, width
        self.thresh = thresh
        self.batch = B = int(batch)        # clips advancing in lock-step, one image of each per frame (BASELINE configs[2])
        fh, fw = -(-height // 16), -(-width // 16)
        dim = cfg.network.DFF_FEAT_DIM
        dev = self.device
        z = lambda *s: torch.zeros(s, device=dev, dtype=torch.float32)
        # static inputs
        self.data = z(B, 3, height, width)
        # frames_by_table (a FramePipeline's segment lanes): cur_segment reads the frames where the caller left them - no staging copy
        self._use_tbl = bool(frames_by_table) and not prefetch
        self.data_tbl = hip.ImageTable(B, 3, height, width, dev).set([self.data[i] for i in range(B)]) if self._use_tbl else None
        self.data_key_old = z(B, 3, height, width)
        self.feat_old = z(B, dim, fh, fw)
        self.mv = z(B, 2, fh, fw)
        self.res = z(B, 3, fh, fw)
        self.im_info = torch.tensor([[height, width, 1.0]] * B, device=dev, dtype=torch.float32)
        R, ncls = cfg.TEST.RPN_POST_NMS_TOP_N, cfg.dataset.NUM_CLASSES
        self._post_full, self.post_bufs = _alloc_post(B, ncls, R, dev)
        self.prefetch = prefetch and cfg.network.add_small_net
        self.data_next = z(B, 3, height, width)        # image of the frame after the current one
        self.small_cur = z(B, dim, fh, fw)             # small-net feature of the current non-key frame
        self.small_next = z(B, dim, fh, fw)
        self.side = streams.new_stream(dev) if self.prefetch else None
        self.feat = None            # the key graph's output feature (static address once captured)
        self.key_graph = self.cur_graph = None
        self.scale = 1.0

    def set_key_feature(self, feat):
        """A lane's copy of the key feature its next frames are served from (on the current stream): channels-last when the lane's frames run
        that way (the copy is the transposition), else a plain copy into feat_shared."""
        if self.feat_shared_cl is not None:
            hip.nchw_to_nhwc(feat, out=self.feat_shared_cl)
        else:
            _stage_inputs([(self.feat_shared, feat)])

    # ---- the two launch sequences -------------------------------------------------------
    def _post(self, out):
        _post_all(out, self._post_full, self.cfg, self.h, self.w, self.scale, self.thresh)

    def _fork_small_next(self):
        """side stream: small-net feature of the next frame (reads data_next, writes small_next)."""
        main = torch.cuda.current_stream(self.device)
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            self.small_next.copy_(self.cur.small_net_feature(self.data_next))

    def _join_small_next(self):
        main = torch.cuda.current_stream(self.device)
        main.wait_stream(self.side)
        self.small_cur.copy_(self.small_next)     # becomes the current frame's feature at the next replay

    def _key_seq(self):
        if self.prefetch:
            self._fork_small_next()
        saved = self.key.taps
        if self.want_taps:
            self.key.taps = self.key_taps = {}
        try:
            out = self.key.forward(data=self.data, im_info=self.im_info, data_key_old=self.data_key_old,
                                   feat_key_old=self.feat_old)
        finally:
            self.key.taps = saved
        self.key_out = out if self.want_taps else None
        self._post(out)
        if self.prefetch:
            self._join_small_next()
        return out['choose_feat_output']

    def _cur_seq(self):
        saved = self.cur.taps
        if self.want_taps:
            self.cur.taps = self.cur_taps = {}
        try:
            if self.prefetch:
                self._fork_small_next()
                out = self.cur.forward(data=self.data, im_info=self.im_info, feat_key=self.feat, motion_vector=self.mv,
                                       res_diff=self.res, small_feat=self.small_cur)
            else:
                out = self.cur.forward(data=self.data_tbl if self._use_tbl else self.data, im_info=self.im_info, feat_key=self.feat,
                                       motion_vector=self.mv, res_diff=self.res, feat_key_cl=self.feat_shared_cl)
        finally:
            self.cur.taps = saved
        self.cur_out = out if self.want_taps else None
        self._post(out)
        if self.prefetch:
            if self.want_taps:
                self.cur_taps['small_feat'] = self.small_cur.clone()   # the join below overwrites small_cur
            self._join_small_next()

    # ---- first frame of a clip (flag 0): eager, no aggregation --------------------------
    def first_frame(self, data, next_data=None):
        """`next_data`: image of the following frame when that frame is a non-key frame and prefetch is
        on — its small-net feature is computed here, like key_frame / cur_frame do for their successor.
        Without it, a cur_frame that follows computes its own feature inline (never a stale one)."""
        ph = torch.zeros((1, self.cfg.network.DFF_FEAT_DIM, 1, 1), device=self.device)
        out = self.key.forward(data=data, im_info=self.im_info, data_key_old=data, feat_key_old=ph)
        self._post(out)
        self._small_valid = False
        if self.prefetch and next_data is not None:
            self.small_cur.copy_(self.cur.small_net_feature(next_data))
            self._small_valid = True
        self.feat_old.copy_(out['choose_feat_output'])
        self.data_key_old.copy_(data)
        if self.feat is not None:
            self.feat.copy_(out['choose_feat_output'])
        else:
            self._first_feat = out['choose_feat_output']
        return self.post_bufs

    def capture(self, warmup=3, key=True, cur=True):
        """Warm up (MIOpen find, workspaces, lazy attributes) on a side stream, then capture."""
        if self.feat_shared is not None:
            key = False
            self.feat = self.feat_shared
        if not self.use_graphs:
            if self.feat_shared is None:
                self.feat = self._first_feat.clone()
            return
        # Warm up and capture on a stream of our own.  torch keeps one BLAS workspace per (handle, stream) and
        # captured GEMMs bake its address in; with torch's default (shared) capture stream — or with two pool
        # streams that are the same hipStream_t (streams.new_stream) — graphs replayed concurrently on different
        # streams (FramePipeline) would share, and corrupt, split-K scratch.
        s = self._capture_stream = streams.new_stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            if key:
                for _ in range(warmup):
                    f = self._key_seq()
                self.feat = f
            if cur:
                for _ in range(warmup):
                    self._cur_seq()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        if key:
            self.key_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.key_graph, stream=self._capture_stream):
                self.feat = self._key_seq()
        if cur:
            self.cur_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.cur_graph, stream=self._capture_stream):
                self._cur_seq()
        if self.feat_shared is None:
            self.feat.copy_(self._first_feat)
        self._small_valid = False      # the warm-up replays left their own small-net feature in small_cur

    def close(self):
        """Drop the captured graphs (and with them their private memory pools), then destroy the streams this object
        created.  The device must be idle with respect to them (call after a synchronize / FramePipeline.join + sync)."""
        self.key_graph = self.cur_graph = None
        for name in ('side', '_capture_stream'):
            st = getattr(self, name, None)
            if st is not None:
                streams.release(st)
                setattr(self, name, None)

    # ---- per-frame entry points ---------------------------------------------------------
    def key_frame(self, data, next_data=None):
        """flag 1: a key frame after the first.  `next_data` = image of the following frame when that
        frame is a non-key frame (its small-net feature is computed alongside).  Returns the
        (dets, counts, keep_idx) device buffers."""
        self.data.copy_(data)
        if self.prefetch:
            self.data_next.copy_(next_data if next_data is not None else data)
        if self.use_graphs:
            self.key_graph.replay()
        else:
            self.feat = self._key_seq()
        self._small_valid = self.prefetch and next_data is not None
        # becomes the "old key" state of the next key frame
        self.feat_old.copy_(self.feat)
        self.data_key_old.copy_(self.data)
        return self.post_bufs

    def cur_frame(self, data, motion_vector, res_diff, next_data=None):
        """flag 2: a non-key frame.  With prefetch on, the small-net feature of THIS frame must have
        been produced by the previous call (pass this frame's image as its `next_data`)."""
        jobs = [(self.data, data), (self.mv, motion_vector), (self.res, res_diff)]
        if self.prefetch:
            if not self._small_valid:
                # nobody computed this frame's small-net feature ahead of time (the previous call had no
                # `next_data`, e.g. a first_frame without it): compute it now instead of using a stale one
                self.small_cur.copy_(self.cur.small_net_feature(data))
            jobs.append((self.data_next, next_data if next_data is not None else data))
        _stage_inputs(jobs)
        if self.use_graphs:
            self.cur_graph.replay()
        else:
            self._cur_seq()
        self._small_valid = self.prefetch and next_data is not None
        return self.post_bufs


    def cur_segment(self, frames):
        """The non-key frames of one segment in ONE pass, batch axis = frames x clips, frame-major (this instance was built with batch =
        len(frames) * B and a shared (B, C, h, w) key feature: every frame of a segment is served from the same key frame, image f * B + b
        from clip b's).  frames: [(data, motion_vector, res_diff), ...], B images each.  Returns the batched (dets, counts, keep_idx)
        buffers: frame f's results are rows [f * B, (f + 1) * B)."""
        B = int(frames[0][0].shape[0])               # clips advancing in lock-step: images per frame; the batch is frame-major (f * B + b)
        if len(frames) * B != self.batch or self.prefetch:
            raise ValueError("cur_segment: this lane takes %d images per pass (and no prefetch)" % self.batch)
        jobs = []
        tbl = self._use_tbl and all(_table_ok(data, self.data.shape[1:], self.device) for data, _, _ in frames)
        for f, (data, mv, res) in enumerate(frames):
            sl = slice(f * B, (f + 1) * B)
            jobs += ([] if tbl else [(self.data[sl], data)]) + [(self.mv[sl], mv), (self.res[sl], res)]
        _stage_inputs(jobs)
        if tbl:
            self.data_tbl.set([data[b] for data, _, _ in frames for b in range(B)])
        elif self._use_tbl:      # frames the table cannot take went through the dense buffer: point the table at it
            self.data_tbl.set([self.data[i] for i in range(self.batch)])
        if self.use_graphs:
            self.cur_graph.replay()
        else:
            self._cur_seq()
        return self._post_full


class KeyBank(object):
    """The image-only half of G consecutive key frames in one pass (batch axis = key frames): backbone of each, FlowNet of each against
    its predecessor.  The late ResNet stages run 38 x 63 maps: one frame's convolutions launch a fraction of a wave of workgroups per
    K slice, several frames' fill the chip (measured per frame: backbone 3150 -> 2080 -> 1860 us, FlowNet 500 -> 273 -> 225 us at G = 1 / 3 / 6,
    profiles/r4/key_batch_probe.txt).  Key frame i of the group then takes conv_feat[i], flow[i], scale[i] for its aggregation."""

    def __init__(self, key_exec, cfg, height, width, device, use_graphs, group, taps=False, batch=1):
        self.key, self.device, self.use_graphs, self.G, self.B = key_exec, device, use_graphs, int(group), int(batch)
        z = lambda *s: torch.zeros(s, device=device, dtype=torch.float32)
        self.data = z(self.G * self.B, 3, height, width)          # group-major: key frame i of the group is images [i * B, (i + 1) * B)
        self.data_old = z(self.G * self.B, 3, height, width)
        # r6: the pass reads its images through pointer tables (hip.ImageTable), i.e. where the caller left them; the dense buffers above are what
        # the tables point at during capture, and the fall-back (LSFA_STAGE_FRAMES=1)
        self.by_table = ZERO_COPY_FRAMES
        n = self.G * self.B
        self.data_tbl = hip.ImageTable(n, 3, height, width, device).set([self.data[i] for i in range(n)]) if self.by_table else None
        self.data_old_tbl = hip.ImageTable(n, 3, height, width, device).set([self.data_old[i] for i in range(n)]) if self.by_table else None
        self.conv_feat = self.flow_out = None
        self.front_graph = self.flow_graph = None
        self.want_taps = taps
        E = torch.cuda.Event
        self.ev_front, self.ev_flow, self.ev_free = E(), E(), E()      # backbone done / FlowNet done / every slice copied out

    def front(self):
        self.conv_feat = self.key.key_backbone(self.data_tbl if self.by_table else self.data)

    def flow(self):
        self.flow_out = self.key.key_flow(self.data_tbl, self.data_old_tbl) if self.by_table else self.key.key_flow(self.data, self.data_old)

    def set_images(self, group, olds):
        """The pass's images: key frame i of the group (B clips each) and its predecessor.  By table: two tiny launches; else the staging copies."""
        nb = self.B
        shape = self.data.shape[1:]
        if self.by_table and all(_table_ok(t, shape, self.device) for t in list(group) + list(olds)):
            self.data_tbl.set([t[b] for t in group for b in range(nb)])
            self.data_old_tbl.set([t[b] for t in olds for b in range(nb)])
            return
        if self.by_table:      # frames the table cannot take: through the dense buffers the tables pointed at when the graphs were captured
            self.data_tbl.set([self.data[i] for i in range(self.G * nb)])
            self.data_old_tbl.set([self.data_old[i] for i in range(self.G * nb)])
        _stage_inputs([(self.data[i * nb:(i + 1) * nb], group[i]) for i in range(self.G)] +
                      [(self.data_old[i * nb:(i + 1) * nb], olds[i]) for i in range(self.G)])

    def capture(self, warmup=2):
        if not self.use_graphs:
            return
        s = self._capture_stream = streams.new_stream(self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.front()
                self.flow()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        self.front_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.front_graph, stream=self._capture_stream):
            self.front()
        self._capture_stream_flow = streams.new_stream(self.device)
        self.flow_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.flow_graph, stream=self._capture_stream_flow):
            self.flow()

    def close(self):
        self.front_graph = self.flow_graph = None
        for name in ('_capture_stream', '_capture_stream_flow'):
            st = getattr(self, name, None)
            if st is not None:
                streams.release(st)
                setattr(self, name, None)

    def run_front(self):
        self.front_graph.replay() if self.use_graphs else self.front()

    def run_flow(self):
        self.flow_graph.replay() if self.use_graphs else self.flow()


class KeyLane(object):
    """Static buffers and the captured parts of a key frame: `front` (backbone) and `flow` (FlowNet) need
    only images; `agg` (flow warp of the previous key feature x scale map, aggregation) produces the
    frame's feature; `tail` (RPN, Proposal, R-FCN head, detection post-processing) consumes it."""

    def __init__(self, key_exec, cfg, height, width, device, thresh, use_graphs, taps=False, batch=1):
        self.key, self.cfg, self.device, self.use_graphs = key_exec, cfg, device, use_graphs
        B = int(batch)
        self.want_taps = taps
        self.taps = {}         # taps=True: stage outputs of this buffer set's last frame (static under replay)
        self.out = None
        self.h, self.w, self.thresh, self.scale = height, width, thresh, 1.0
        fh, fw = -(-height // 16), -(-width // 16)
        z = lambda *s: torch.zeros(s, device=device, dtype=torch.float32)
        self.data = z(B, 3, height, width)
        self.data_key_old = z(B, 3, height, width)
        self.feat_old = z(B, cfg.network.DFF_FEAT_DIM, fh, fw)
        self.im_info = torch.tensor([[height, width, 1.0]] * B, device=device, dtype=torch.float32)
        R, ncls = cfg.TEST.RPN_POST_NMS_TOP_N, cfg.dataset.NUM_CLASSES
        self._post_full, self.post_bufs = _alloc_post(B, ncls, R, device)
        self.conv_feat = self.flow_out = None
        self.flow_graph = None
        self.feat = None
        self.front_graph = self.agg_graph = self.tail_graph = None
        self.ev_front, self.ev_flow = torch.cuda.Event(), torch.cuda.Event()

    def _tapped(self, fn):
        """Run one part with the executor's tap dict pointed at this lane's (parity tests only)."""
        if not self.want_taps:
            return fn()
        saved = self.key.taps
        self.key.taps = self.taps
        try:
            return fn()
        finally:
            self.key.taps = saved

    def front(self):
        self.conv_feat = self._tapped(lambda: self.key.key_backbone(self.data))

    def flow(self):
        self.flow_out = self.key.key_flow(self.data, self.data_key_old)

    def agg(self):
        self.feat = self._tapped(lambda: self.key.key_aggregate(self.conv_feat, self.flow_out[0], self.flow_out[1],
                                                                self.feat_old))

    def tail(self):
        cfg = self.cfg
        out = self._tapped(lambda: self.key.key_heads(self.feat, self.im_info))
        self.out = out if self.want_taps else None
        _post_all(out, self._post_full, cfg, self.h, self.w, self.scale, self.thresh)

    def capture(self, warmup=3):
        if not self.use_graphs:
            return
        s = self._capture_stream = streams.new_stream(self.device)     # see FrameGraphs.capture
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.front()
                self.flow()
                self.agg()
                self.tail()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        self.front_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.front_graph, stream=self._capture_stream):
            self.front()
        # FlowNet replays on another stream, beside the backbone: its own capture stream (BLAS workspace)
        self._capture_stream_flow = streams.new_stream(self.device)
        self.flow_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.flow_graph, stream=self._capture_stream_flow):
            self.flow()
        self.agg_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.agg_graph, stream=self._capture_stream):
            self.agg()
        # the tail replays where FlowNet does (after it, same stream), so it may share that capture stream
        self.tail_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.tail_graph, stream=self._capture_stream_flow):
            self.tail()

    def close(self):
        self.front_graph = self.flow_graph = self.agg_graph = self.tail_graph = None
        for name in ('_capture_stream', '_capture_stream_flow'):
            st = getattr(self, name, None)
            if st is not None:
                streams.release(st)
                setattr(self, name, None)

    def run_front(self):
        self.front_graph.replay() if self.use_graphs else self.front()

    def run_flow(self):
        self.flow_graph.replay() if self.use_graphs else self.flow()

    def run_agg(self):
        self.agg_graph.replay() if self.use_graphs else self.agg()

    def run_tail(self):
        self.tail_graph.replay() if self.use_graphs else self.tail()


class FramePipeline(object):
    """One clip, frames handed over in display order, pipelined over HIP streams.

    Data dependencies of the frame loop (dff_rfcn/core/tester.py:237-281): a non-key frame needs the
    feature of the latest key frame and its own image / motion vectors / residual — nothing from the
    neighbouring non-key frames.  A key frame needs the previous key frame's IMAGE for its backbone +
    FlowNet part (the bulk of it) and the previous key frame's FEATURE only for the warp /
    aggregation at its end; it needs nothing from the non-key frames in between.  The reference runs
    everything serially (one executor, a blocking .asnumpy() per frame).  Here:

      * a key frame is four captured graphs: `front` (backbone) and `agg` (flow warp, aggregation ->
        the frame's feature) on the key stream, `flow` (FlowNet) and `tail` (RPN, Proposal, R-FCN head,
        detections) on a second stream: FlowNet runs beside the backbone, and the next key frame's
        backbone starts as soon as the feature exists, beside this frame's single-workgroup tail.
        Two sets of key-frame buffers alternate, so a key frame never overwrites the feature the
        previous segment's non-key frames are still being served from;
      * the non-key frames of a segment alternate over `lanes` streams, each lane with its own captured
        graph and static buffers, all reading one shared copy of the key feature ("hand-over": one
        10 MB copy per segment, after the key frame's `agg` and after every lane has finished the
        previous segment);
      * issue order: by default frames are queued as they are handed over (display order).
        `lookahead=True` queues key frame k+1 BEFORE the non-key frames that precede it: `cur_frame`
        then only records the frame and the segment is queued when the next key frame (or `flush` /
        `join` / `first_frame`) arrives, so non-key detections are delivered one segment late.  A
        graph launch blocks the host while the previous launch of the same graph is still running, and
        under event instrumentation the key stream was seen idling behind the host; un-instrumented the
        two orders measure the same (1324 vs 1325 frames/s), hence the simpler default.

    What this buys on a 256-CU part: every frame ends with work that occupies one or a few CUs
    (Proposal's single workgroup, the R-FCN head, the detection NMS: ~40 % of a non-key frame's
    time) and the key frame's late ResNet stages launch grids well under 256 workgroups; with
    independent frames in flight those CUs run another frame's convolutions instead of idling.
    Results are those of the serial loop: same launch sequences, same inputs, no shared scratch.
    Streams: core/streams.py probes for streams on distinct hardware queues (the runtime has 4;
    streams sharing one run strictly in turn): key, FlowNet/tail and two lanes get one each; further
    lanes share the FlowNet stream's queue.
    The caller must keep each key frame's `data` tensor unmodified until the next key frame has
    been queued (it is read again as that frame's `data_key_old`).  r6: batched passes (segment > 0, key_group > 1) read the frames
    WHERE THE CALLER LEFT THEM (hip.ImageTable: no staging copy), so those tensors must stay unmodified until the pass has run - i.e.
    until their results have been delivered; dropping the Python reference is fine (the pipeline holds one and calls record_stream),
    overwriting the storage in place is not.  LSFA_STAGE_FRAMES=1 restores the staging copies.
    """

    def __init__(self, key_exec, cur_exec, cfg, height, width, device, thresh=1e-4, use_graphs=True, lanes=2,
                 flow_stream=True, lookahead=False, taps=False, batch=1, layout=None, segment=0, key_group=1, ramp=False):
        """batch > 1: that many clips advance in lock-step — every tensor handed to first_frame / key_frame /
        cur_frame carries one image (motion-vector field, residual) per clip on its batch axis, and the
        detection buffers gain a leading clip axis.
        layout (default: $LSFA_STREAM_LAYOUT or 'probe'): how the work streams are picked.  'probe' times pairs of
        candidate streams (core/streams.py) and keeps mutually concurrent ones — a speed matter only, but the probe
        is perturbed by whatever else runs on the GPU, so the outcome is logged in `layout_used` and the other
        values force one: 'plain' = a fresh stream per role, no probing; 'one-queue' = what the probe returns when
        every candidate looks aliased to the key stream (no FlowNet stream, lanes on spare streams).  Results are
        the same under every layout (tests/test_graph_gpu.py runs the pipeline under each)."""
        dev = torch.device(device)
        self.batch = B = int(batch)
        self.segment, self.key_group = int(segment), int(key_group)
        if self.segment > 0 and lanes < 2:
            raise ValueError("FramePipeline: segment batching alternates over two lane streams (lanes >= 2)")
        self.device, self.cfg, self.key_exec = dev, cfg, key_exec
        self.h, self.w, self.thresh, self.scale = height, width, thresh, 1.0
        self.lookahead = lookahead
        fh, fw = -(-height // 16), -(-width // 16)
        dim = cfg.network.DFF_FEAT_DIM
        self.feat_cur = torch.zeros((B, dim, fh, fw), device=dev, dtype=torch.float32)   # what the non-key lanes read
        self.feat0 = torch.zeros((B, dim, fh, fw), device=dev, dtype=torch.float32)      # feature of a clip's frame 0
        self.klanes = [KeyLane(key_exec, cfg, height, width, dev, thresh, use_graphs, taps, B) for _ in range(2)]
        self.lanes = [FrameGraphs(key_exec, cur_exec, cfg, height, width, dev, thresh, use_graphs, prefetch=False,
                                  feat_shared=self.feat_cur, taps=taps, batch=B) for _ in range(lanes)]
        for ln in self.lanes[1:]:
            ln.feat_shared_cl = self.lanes[0].feat_shared_cl       # one shared copy, like feat_cur
        # segment > 0: the non-key frames of a segment (all served from one key feature) go through the network in ONE pass of `segment`
        # frames on the batch axis - what the reference's own batch test symbol does (get_batch_test_symbol,
        # resnet_v1_101_flownet_rfcn.py:661-751) - on `lanes` (>= 2) alternating lanes, each with its own copy of the key feature; a shorter run of
        # non-key frames (the end of a clip) takes the per-frame lanes.  key_group > 1: see KeyBank and key_frame(upcoming=...).
        self.feat_seg = [torch.zeros((B, dim, fh, fw), device=dev, dtype=torch.float32) for _ in range(max(2, lanes) if self.segment else 0)]
        self.seg_lanes = [FrameGraphs(key_exec, cur_exec, cfg, height, width, dev, thresh, use_graphs, prefetch=False,
                                      feat_shared=f, taps=taps, batch=self.segment * B, frames_by_table=ZERO_COPY_FRAMES) for f in self.feat_seg]
        # one bank per group size 2 .. min(key_group, 6) and the full size: the tail of a run of key frames (fewer images ahead than key_group - 1)
        # is a smaller group - with key_group > 6 the largest size there is a bank for (a bank per size would hold key_group^2 / 2 images' maps)
        # (two of the full size, alternating: the next group's pass may start while this group's key frames still take their slices)
        self.banks = {g: [KeyBank(key_exec, cfg, height, width, dev, use_graphs, g, taps, B) for _ in range(2 if g == self.key_group else 1)]
                      for g in sorted(set(range(2, min(self.key_group, 6) + 1)) | ({self.key_group} if self.key_group >= 2 else set()))}
        self._bank_turn = 0
        self._bank_ready = []                        # [(bank, slot, image tensor)]: fronts of upcoming key frames already computed (the tensor is HELD: its
                                                     # storage cannot be freed and handed to another image at the same address while its front waits)
        # ramp (True = (1, 2)): while the pipeline is empty (the first key frames of a clip, or after flush() / join()) nothing overlaps a
        # pass of key_group fronts and every lane waits for it: with a ramp the first pass after that is one front, the second a group of
        # two, then full groups - the first detections arrive after 3 ms instead of 10.  Off by default since the end of r4: small passes
        # cost 1.4-1.8x per frame what a pass of six does, and over a 20-interval region the ramp lost 4-6 % (3063 vs 3261 frames/s).
        # (ramp may also be the tuple of those first pass sizes; $LSFA_RAMP = "2" / "1,2" / "" overrides it: lab)
        steps = (1, 2) if ramp is True else tuple(int(v) for v in ramp) if ramp else ()
        if os.environ.get('LSFA_RAMP') is not None:
            steps = tuple(int(v) for v in os.environ['LSFA_RAMP'].split(',') if v.strip())
        self.ramp_steps = tuple(min(max(v, 1), self.key_group) for v in steps) if self.key_group > 2 else ()
        self.ramp = bool(self.ramp_steps)
        self._ramp_step = 0
        self.group_sizes = []                        # the sizes of the passes issued so far (diagnostics / tests)
        self._next_seg = 0
        want = 1 + (1 if flow_stream else 0) + lanes
        layout = layout or os.environ.get('LSFA_STREAM_LAYOUT', 'probe')
        if layout == 'probe':
            chosen, aliased = streams.concurrent_streams(want, dev)
        elif layout == 'plain':
            # LSFA_STREAM_PRIO (experiments): which of the streams are created with the device's greatest priority
            prio = os.environ.get('LSFA_STREAM_PRIO', '')
            roles = ['key'] + (['flow'] if flow_stream else []) + ['lanes'] * lanes
            chosen, aliased = [streams.new_stream(dev, high_priority=(r in prio.split(','))) for r in roles], []
        elif layout == 'one-queue':
            chosen, aliased = [streams.new_stream(dev)], [(streams.new_stream(dev), 0) for _ in range(want)]
        else:
            raise ValueError("FramePipeline: layout must be 'probe', 'plain' or 'one-queue', got %r" % (layout,))
        self.hw_queues = len(chosen)
        self._owned_streams = list(chosen) + [st for st, _ in aliased]
        self.s_key = chosen[0]
        rest = chosen[1:]
        self.s_flow = rest.pop(0) if (flow_stream and rest) else None
        flow_q = chosen.index(self.s_flow) if self.s_flow is not None else -1
        spare = [st for st, q in aliased if q == flow_q] + [st for st, q in aliased if q not in (flow_q, 0)] + \
                [st for st, q in aliased if q == 0]
        self.s_lane = [rest.pop(0) if rest else (spare.pop(0) if spare else streams.new_stream(dev))
                       for _ in range(lanes)]
        self.layout_used = '%s: %d concurrent stream(s) of %d wanted, FlowNet stream %s, lanes %s' % (
            layout, len(chosen), want, 'own' if self.s_flow is not None else 'none (key stream)',
            ['own' if st in chosen else 'spare' for st in self.s_lane])
        if os.environ.get('LSFA_LOG_LAYOUT') == '1':
            sys.stderr.write('[lsfa] FramePipeline %dx%d streams: %s\n' % (height, width, self.layout_used))
        E = torch.cuda.Event
        self.ev_in, self.ev_flow, self.ev_tail, self.ev_handover = E(), E(), E(), E()
        # A key frame's feature leaves its key lane right after `agg`: into one of key_group + 2 pool buffers, which is what the next key
        # frame's aggregation, the non-key lanes' copies and the caller's `feat` read.  The key lane (two alternate) is then only held by
        # its own tail; with the lanes copying out of the key lane itself the aggregations of a group advanced at the pace of the
        # SEGMENTS (buffer k + 2 waited for segment k's copy, which waited for its lane to finish the previous group's segments) and the
        # next pass of fronts started ~4 ms late (profiles/r4/pipeline_timeline.txt: the key queue idle a third of the time).
        self._feat_pool = [torch.zeros((B, dim, fh, fw), device=dev, dtype=torch.float32) for _ in range(self.key_group + 2)]
        self.ev_feat = [E() for _ in self._feat_pool]      # pool buffer i holds its key frame's feature
        self.ev_tail_of = [E(), E()]              # key lane i's tail is done reading its feature
        self.ev_lane = [E() for _ in range(lanes)]
        self.captured = False
        self.delivering = None
        self._next = self._nkey = 0
        self._feat_latest = self._prev_key_data = None
        self._seg_feat = self._seg_event = None     # feature (and its event) the buffered segment is served from
        # which key frame's feature the non-key lanes' copies hold (feat_cur, feat_seg[i]) against the one the current segment needs;
        # _handed[b]: events of the copies taken from key buffer b's current feature (key frame k + 2 overwrites it after them)
        self._seg_key, self._seg_buf, self._cur_key, self._lane_key = 0, None, 0, [-1] * len(self.seg_lanes)
        self._handed = [[] for _ in self._feat_pool]
        self._seg_needs_handover = False
        self._held = []                              # non-key frames recorded but not yet queued

    # the state the serial FrameGraphs exposes under the same names
    @property
    def feat(self):
        return self._feat_latest

    @property
    def feat_old(self):
        return self._feat_latest

    @property
    def data_key_old(self):
        return self._prev_key_data

    def set_scale(self, im_scale):
        self.scale = float(im_scale)
        for g in self.klanes + self.lanes + self.seg_lanes:
            g.scale = float(im_scale)
            g.im_info[:, 2] = float(im_scale)

    def _all_streams(self):
        return [self.s_key] + self.s_lane + ([self.s_flow] if self.s_flow is not None else [])

    def close(self):
        """Release what the pipeline owns on the device side: drains it, drops the captured graphs of every lane (with their
        private memory pools — the bulk of what a process that builds a pipeline per frame shape would accumulate) and
        hands its streams back to core/streams.py (parked there: see streams._REUSE for why they are not recycled)."""
        self.join()
        torch.cuda.synchronize(self.device)
        for g in self.klanes + self.lanes + self.seg_lanes + [b for bs in self.banks.values() for b in bs]:
            g.close()
        extra = [st for st in self.s_lane if st not in self._owned_streams]
        for st in self._owned_streams + extra:
            streams.release(st)
        self._owned_streams = []
        self.captured = False

    def drop_fronts(self):
        """Forget the fronts the bank computed ahead for key frames that will not come (the caller abandons the run of frames it announced)."""
        # the abandoned group never reaches its last slot, where `ev_free` is normally recorded: record it here, on the stream the slices are
        # taken on (in order behind the slice copies and the FlowNet pass already queued there), so that the next pass that reuses such a bank
        # waits for THIS group's readers instead of an older event
        sa = self.s_flow if self.s_flow is not None else self.s_key
        for bank in {id(b): b for b, _, _ in self._bank_ready}.values():
            sa.wait_event(bank.ev_flow)
            sa.wait_event(bank.ev_front)
            bank.ev_free.record(sa)
        self._bank_ready = []
        self._ramp_step = 0

    def flush(self):
        """Queue the non-key frames recorded so far (end of a clip, or before reading results: the caller is about to let the pipeline
        run empty, so the next fresh pass of key fronts starts the ramp again)."""
        self._issue_segment()
        if not self._bank_ready:
            self._ramp_step = 0

    def join(self):
        """Everything handed over so far is queued, and the caller's stream waits for it."""
        self.flush()
        main = torch.cuda.current_stream(self.device)
        for s in self._all_streams():
            main.wait_stream(s)

    def first_frame(self, data):
        """flag 0 (first frame of a clip): drains the pipeline, runs eagerly on the caller's stream."""
        self.join()
        self._next = self._nkey = self._next_seg = 0          # the lane / buffer of a frame depends only on its position in the clip
        self._bank_ready = []
        self._ramp_step = 0
        lane, cfg = self.klanes[0], self.cfg
        saved = self.key_exec.taps
        if lane.want_taps:
            self.key_exec.taps = self.first_taps = {}
        try:
            conv_feat, _, _ = self.key_exec.key_front(data, None)
            out = self.key_exec.key_back(conv_feat, None, None, None, lane.im_info)
        finally:
            self.key_exec.taps = saved
        self.first_out = out if lane.want_taps else None
        if not hasattr(self, '_first_post'):
            self._first_full, self._first_post = _alloc_post(self.batch, cfg.dataset.NUM_CLASSES, cfg.TEST.RPN_POST_NMS_TOP_N,
                                                             self.device)
        _post_all(out, self._first_full, cfg, self.h, self.w, self.scale, self.thresh)
        self.feat0.copy_(out['choose_feat_output'])
        self._feat_latest, self._prev_key_data = self.feat0, data
        self._publish_from_main()
        return self._first_post

    def _set_lanes_feature(self, feat):
        """the per-frame lanes share ONE copy of the key feature (feat_cur; channels-last: the first lane's buffer, which the others alias)"""
        self.lanes[0].set_key_feature(feat) if self.lanes else _stage_inputs([(self.feat_cur, feat)])

    def _publish_from_main(self):
        """The clip's first feature goes to the lanes directly; every stream and event starts from here."""
        main = torch.cuda.current_stream(self.device)
        self._set_lanes_feature(self._feat_latest)
        for s in self._all_streams():
            s.wait_stream(main)
        for e in [self.ev_handover, self.ev_tail] + self.ev_feat + self.ev_lane + self.ev_tail_of:
            e.record(main)
        self._seg_feat, self._seg_event, self._seg_needs_handover = self._feat_latest, None, False
        self._seg_key += 1
        self._seg_buf, self._cur_key, self._lane_key, self._handed = None, self._seg_key, [-1] * len(self.seg_lanes), [[] for _ in self._feat_pool]

    def capture(self, warmup=3):
        for lane in self.klanes:
            lane.feat_old.copy_(self.feat0)
            lane.capture(warmup)
        for lane in self.lanes + self.seg_lanes:
            lane.capture(warmup, key=False)
        for bs in self.banks.values():
            for bank in bs:
                bank.capture()
        torch.cuda.synchronize(self.device)
        self.captured = True
        self._publish_from_main()

    def key_frame(self, data, deliver=None, ready=None, upcoming=None):
        """flag 1.  `deliver(bufs)` is called with the stream of the frame's tail current, right after
        the frame is queued; use it to queue copies of the (dets, counts, keep_idx) buffers.  The
        inputs must be complete on the device, or `ready` an event recorded after the work that
        produces them (the caller's stream is deliberately NOT waited on: it would serialise the
        pipeline).  With lookahead, the non-key frames recorded since the previous key frame are
        queued right after this frame.
        upcoming (key_group = G > 1): the images of the next G - 1 key frames of the clip, already on the device.  When the bank holds no
        front for this frame, the fronts (backbone, FlowNet) of this frame and those G - 1 are computed in one pass (KeyBank) and the
        next G - 1 key_frame calls - which must hand over exactly those tensors - take theirs from the bank.  With fewer than G - 1
        images (the end of a clip) the group is that much smaller; without any the frame's front is computed alone, as with key_group = 1."""
        bank, slot, group = None, -1, None
        # checked before ANYTHING is queued or counted: the call leaves no trace.  The front belongs to the tensor handed over in `upcoming` (held by
        # the bank) or to a view of the same storage at the same offset and shape.
        if self._bank_ready and not (self._bank_ready[0][2] is data or (self._bank_ready[0][2].data_ptr() == data.data_ptr()
                                                                       and self._bank_ready[0][2].shape == data.shape)):
            raise ValueError("FramePipeline.key_frame: the bank holds the front of another image (hand the tensors of `upcoming` over in "
                             "order, or drop_fronts())")
        if not self.lookahead:
            self._issue_segment()
        if self._bank_ready:
            bank, slot, _ = self._bank_ready.pop(0)
        b = self._nkey % 2
        self._nkey += 1
        lane, s = self.klanes[b], self.s_key
        if bank is None:
            cap = self.ramp_steps[self._ramp_step] if self._ramp_step < len(self.ramp_steps) else self.key_group
            self._ramp_step += 1
            g = min(cap, 1 + len(upcoming or ())) if self.banks else 1
            g = max([n for n in self.banks if n <= g] or [1])         # (key_group > 6: not every size has a bank)
            self.group_sizes.append(g)
            del self.group_sizes[:-64]
            if g >= 2:
                self._bank_turn += 1
                bank, group, slot = self.banks[g][self._bank_turn % len(self.banks[g])], [data] + list(upcoming[:g - 1]), 0
                self._bank_ready = [(bank, i, group[i]) for i in range(1, g)]
        # ---- the image-only part, when this call starts one: a pass of the bank (or this frame's front alone) on the key stream, FlowNet
        #      beside it; nothing here waits for aggregations or tails, so passes run back to back
        src = bank if slot >= 0 else lane
        if slot < 0 or group is not None:
            with torch.cuda.stream(s):
                if ready is not None:
                    s.wait_event(ready)
                if slot < 0:
                    # the key lane's own front: its static buffers are free once its previous frame (k - 2) is through tail and `deliver`
                    s.wait_event(self.ev_tail_of[b])
                    for t in (data, self._prev_key_data):
                        t.record_stream(s)               # the caller may drop its reference right after this call
                    _stage_inputs([(lane.data, data), (lane.data_key_old, self._prev_key_data)])
                else:
                    s.wait_event(bank.ev_free)           # the bank's previous group has taken all its slices
                    # frame i of the group against its predecessor (the previous key frame for i = 0)
                    olds = [self._prev_key_data] + group[:-1]
                    for t in group + olds[:1]:
                        t.record_stream(s)
                    bank.set_images(group, olds)
                if self.s_flow is not None:
                    self.ev_in.record(s)
                    with torch.cuda.stream(self.s_flow):
                        self.s_flow.wait_event(self.ev_in)
                        src.run_flow()
                        src.ev_flow.record(self.s_flow)
                    src.run_front()
                    src.ev_front.record(s)
                else:
                    src.run_front()
                    src.run_flow()
                    src.ev_front.record(s)
                    src.ev_flow.record(s)
        # ---- this key frame's own part: its slice of the bank, aggregation, the feature into the pool - on the FlowNet / tail stream when
        #      there is one (in order behind the FlowNet pass and the previous key frames' tails), so that the key stream is free for the next pass
        sa = self.s_flow if self.s_flow is not None else s
        with torch.cuda.stream(sa):
            if ready is not None:
                sa.wait_event(ready)
            sa.wait_event(src.ev_front)
            sa.wait_event(src.ev_flow)
            if not lane.use_graphs:
                # eager mode: the maps were allocated on the streams that produced them and are read here
                for t in (src.conv_feat,) + tuple(src.flow_out):
                    t.record_stream(sa)
            # this key lane's previous frame (k - 2) is through its tail and its `deliver`: the heads are done reading its feature, and whoever
            # reads the lane's taps there (tests) is done with the static buffers that the copies from the bank / `agg` overwrite
            sa.wait_event(self.ev_tail_of[b])
            if slot >= 0:
                # this frame's slice of the bank -> what the key lane's `agg` reads (static buffers under replay)
                sl = slice(slot * bank.B, (slot + 1) * bank.B)
                parts = (bank.conv_feat[sl], bank.flow_out[0][sl], bank.flow_out[1][sl])
                # (one lsfa_copy_many launch together with the previous key frame's feature - queued earlier on this stream: it exists)
                if lane.use_graphs:
                    _stage_inputs([(lane.conv_feat, parts[0]), (lane.flow_out[0], parts[1]), (lane.flow_out[1], parts[2]),
                                   (lane.feat_old, self._feat_latest)])
                else:
                    lane.conv_feat, lane.flow_out = parts[0], (parts[1], parts[2])
                    _stage_inputs([(lane.feat_old, self._feat_latest)])
                if slot == bank.G - 1:
                    bank.ev_free.record(sa)
                if lane.want_taps:
                    lane.taps['backbone_feat'] = lane.conv_feat
            else:
                _stage_inputs([(lane.feat_old, self._feat_latest)])   # the previous key frame's feature
            lane.run_agg()
            # out of the key lane: pool buffer j, once the copies taken from its previous occupant (key frame k - len(pool)) are done
            j = (self._nkey - 1) % len(self._feat_pool)
            for e in self._handed[j]:
                sa.wait_event(e)
            self._handed[j] = []
            pooled = self._feat_pool[j]
            _stage_inputs([(pooled, lane.feat)])
            self.ev_feat[j].record(sa)
            lane.run_tail()                      # heads + detections
            if deliver is not None:
                self.delivering = lane           # whose buffers `deliver` sees (tests read its taps)
                deliver(lane.post_bufs)
            self.ev_tail.record(sa)
            self.ev_tail_of[b].record(sa)
        if self.lookahead:
            self._issue_segment()                # the frames BEFORE this key frame, served from the previous feature
        self._feat_latest, self._prev_key_data = pooled, data
        self._seg_feat, self._seg_event, self._seg_needs_handover, self._seg_buf = pooled, self.ev_feat[j], True, j
        self._seg_key += 1
        return lane.post_bufs

    def cur_frame(self, data, motion_vector, res_diff, deliver=None, ready=None):
        """flag 2.  Recorded; queued on the next lane when the segment is issued (see the class
        docstring).  `deliver(bufs)` is called then, with the lane's stream current; the buffers are
        valid until the lane's next frame, so copy them out there."""
        self._held.append((data, motion_vector, res_diff, deliver, ready))
        if not self.lookahead and (self.segment == 0 or len(self._held) >= self.segment):
            self._issue_segment()

    def _copied_out(self, s):
        """a copy of the current segment's key feature has been queued on `s`: the key buffer it came from waits for it before its next use"""
        if self._seg_buf is not None:
            e = torch.cuda.Event()
            e.record(s)
            self._handed[self._seg_buf].append(e)

    def stream_of_next_segment(self):
        """The stream the segment whose frames are handed over NEXT will be queued on (batched segments alternate over the lanes; a segment still
        held goes first): where a caller should queue the upload of that segment's inputs, so that the copy does not sit behind the OTHER lane's
        pass (bench.py: the lanes' parity against the caller's own interval count made the upload mode read 7-12 % low when it started on an odd
        count)."""
        if self.segment > 0:
            return self.s_lane[(self._next_seg + (1 if self._held else 0)) % len(self.seg_lanes)]
        return self.s_lane[self._next % len(self.lanes)]

    def _issue_batched(self):
        """The held frames (exactly `segment` of them, all of one segment) as one pass on the next segment lane."""
        i = self._next_seg
        self._next_seg = (i + 1) % len(self.seg_lanes)
        lane, s = self.seg_lanes[i], self.s_lane[i]
        with torch.cuda.stream(s):
            for _, _, _, _, ready in self._held:
                if ready is not None:
                    s.wait_event(ready)
            if self._lane_key[i] != self._seg_key:
                # this lane's copy of the segment's key feature (the lane's previous segment is done with the old one: stream order)
                if self._seg_event is not None:
                    s.wait_event(self._seg_event)
                lane.set_key_feature(self._seg_feat)
                self._copied_out(s)
                self._lane_key[i] = self._seg_key
            frames = []
            for data, motion_vector, res_diff, _, _ in self._held:
                for t in (data, motion_vector, res_diff):
                    t.record_stream(s)
                frames.append((data, motion_vector, res_diff))
            full = lane.cur_segment(frames)
            self.delivering = lane
            nb = self.batch
            for f, (_, _, _, deliver, _) in enumerate(self._held):
                if deliver is not None:
                    bufs = tuple(t[f] if nb == 1 else t[f * nb:(f + 1) * nb] for t in full)     # like post_bufs: no clip axis for one clip
                    bufs[0].lsfa_segment = (full[0].lsfa_flat, f, len(self._held))      # a consumer may take the whole segment with one copy
                    deliver(bufs)
            self.ev_lane[i].record(s)
        self._held = []

    def _issue_segment(self):
        if not self._held:
            return
        if self.segment > 0 and len(self._held) == self.segment:
            return self._issue_batched()
        if self._cur_key != self._seg_key:
            s = self.s_lane[0]
            with torch.cuda.stream(s):
                if self._seg_event is not None:
                    s.wait_event(self._seg_event)        # the segment's key feature exists
                for e in self.ev_lane[1:]:
                    s.wait_event(e)                      # every lane has finished the previous segment
                self._set_lanes_feature(self._seg_feat)
                self.ev_handover.record(s)
                self._copied_out(s)
            self._cur_key = self._seg_key
            self._seg_needs_handover = False
        for data, motion_vector, res_diff, deliver, ready in self._held:
            i = self._next
            self._next = (i + 1) % len(self.lanes)
            s = self.s_lane[i]
            with torch.cuda.stream(s):
                if ready is not None:
                    s.wait_event(ready)
                s.wait_event(self.ev_handover)
                for t in (data, motion_vector, res_diff):
                    t.record_stream(s)
                bufs = self.lanes[i].cur_frame(data, motion_vector, res_diff)
                if deliver is not None:
                    self.delivering = self.lanes[i]
                    deliver(bufs)
                self.ev_lane[i].record(s)
        self._held = []
