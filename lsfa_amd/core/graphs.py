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
import torch

from lsfa_amd import hip


class FrameGraphs(object):
    def __init__(self, key_exec, cur_exec, cfg, height, width, device, thresh=1e-4, use_graphs=True, prefetch=True):
        self.key, self.cur, self.cfg = key_exec, cur_exec, cfg
        self.device = torch.device(device)
        self.use_graphs = use_graphs
        self.h, self.w = height, width
        self.thresh = thresh
        fh, fw = -(-height // 16), -(-width // 16)
        dim = cfg.network.DFF_FEAT_DIM
        dev = self.device
        z = lambda *s: torch.zeros(s, device=dev, dtype=torch.float32)
        # static inputs
        self.data = z(1, 3, height, width)
        self.data_key_old = z(1, 3, height, width)
        self.feat_old = z(1, dim, fh, fw)
        self.mv = z(1, 2, fh, fw)
        self.res = z(1, 3, fh, fw)
        self.im_info = torch.tensor([[height, width, 1.0]], device=dev, dtype=torch.float32)
        R, ncls = cfg.TEST.RPN_POST_NMS_TOP_N, cfg.dataset.NUM_CLASSES
        self.post_bufs = (torch.zeros((ncls, R, 5), dtype=torch.float64, device=dev),
                          torch.zeros(ncls, dtype=torch.int32, device=dev),
                          torch.full((ncls, R), -1, dtype=torch.int32, device=dev))
        self.prefetch = prefetch and cfg.network.add_small_net
        self.data_next = z(1, 3, height, width)        # image of the frame after the current one
        self.small_cur = z(1, dim, fh, fw)             # small-net feature of the current non-key frame
        self.small_next = z(1, dim, fh, fw)
        self.side = torch.cuda.Stream(device=dev) if self.prefetch else None
        self.feat = None            # the key graph's output feature (static address once captured)
        self.key_graph = self.cur_graph = None
        self.scale = 1.0

    # ---- the two launch sequences -------------------------------------------------------
    def _post(self, out):
        cfg = self.cfg
        return hip.det_postprocess(out['rois_output'], out['bbox_pred_reshape_output'][0], out['cls_prob_reshape_output'][0],
                                   self.h, self.w, self.scale, score_thresh=self.thresh, nms_thresh=cfg.TEST.NMS,
                                   max_per_image=cfg.TEST.max_per_image, class_agnostic=cfg.CLASS_AGNOSTIC,
                                   out=self.post_bufs)

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
        out = self.key.forward(data=self.data, im_info=self.im_info, data_key_old=self.data_key_old,
                               feat_key_old=self.feat_old)
        self._post(out)
        if self.prefetch:
            self._join_small_next()
        return out['choose_feat_output']

    def _cur_seq(self):
        if self.prefetch:
            self._fork_small_next()
            out = self.cur.forward(data=self.data, im_info=self.im_info, feat_key=self.feat, motion_vector=self.mv,
                                   res_diff=self.res, small_feat=self.small_cur)
        else:
            out = self.cur.forward(data=self.data, im_info=self.im_info, feat_key=self.feat, motion_vector=self.mv,
                                   res_diff=self.res)
        self._post(out)
        if self.prefetch:
            self._join_small_next()

    # ---- first frame of a clip (flag 0): eager, no aggregation --------------------------
    def first_frame(self, data):
        ph = torch.zeros((1, self.cfg.network.DFF_FEAT_DIM, 1, 1), device=self.device)
        out = self.key.forward(data=data, im_info=self.im_info, data_key_old=data, feat_key_old=ph)
        self._post(out)
        self.feat_old.copy_(out['choose_feat_output'])
        self.data_key_old.copy_(data)
        if self.feat is not None:
            self.feat.copy_(out['choose_feat_output'])
        else:
            self._first_feat = out['choose_feat_output']
        return self.post_bufs

    def capture(self, warmup=3):
        """Warm up (MIOpen find, workspaces, lazy attributes) on a side stream, then capture."""
        if not self.use_graphs:
            self.feat = self._first_feat.clone()
            return
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                f = self._key_seq()
            self.feat = f
            for _ in range(warmup):
                self._cur_seq()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        self.key_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.key_graph):
            self.feat = self._key_seq()
        self.cur_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.cur_graph):
            self._cur_seq()
        self.feat.copy_(self._first_feat)

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
        # becomes the "old key" state of the next key frame
        self.feat_old.copy_(self.feat)
        self.data_key_old.copy_(self.data)
        return self.post_bufs

    def cur_frame(self, data, motion_vector, res_diff, next_data=None):
        """flag 2: a non-key frame.  With prefetch on, the small-net feature of THIS frame must have
        been produced by the previous call (pass this frame's image as its `next_data`)."""
        self.data.copy_(data)
        if self.prefetch:
            self.data_next.copy_(next_data if next_data is not None else data)
        self.mv.copy_(motion_vector)
        self.res.copy_(res_diff)
        if self.use_graphs:
            self.cur_graph.replay()
        else:
            self._cur_seq()
        return self.post_bufs
