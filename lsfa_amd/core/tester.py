"""Predictor, im_detect and the per-frame evaluation loop.

Mirror of dff_rfcn/core/tester.py: `Predictor(symbol, data_names, label_names, context,
max_data_shapes, provide_data, provide_label, arg_params, aux_params)` (:28-41),
`im_detect(predictor, data_batch, data_names, scales, cfg)` (:130-160) and `pred_eval(...)`
(:192-299) keep the reference's signatures, return values and output names.

What is different by design: nothing is copied to the host inside the frame loop.  The
reference does .asnumpy() on rois / cls_prob / bbox_pred every frame, decodes boxes in numpy
and runs 30 numpy NMS calls (tester.py:138-152, :265-281); here `im_detect_device` leaves the
network outputs on the device and one HIP launch does decode + clip + per-class NMS + the
max_per_image cap (lsfa_det_postprocess).  `im_detect` still exists with the reference's
host-side return types for callers that want them.
"""
import os
import time

import numpy as np
import torch

from lsfa_amd import hip


class Predictor(object):
    def __init__(self, symbol, data_names, label_names, context='cuda:0', max_data_shapes=None, provide_data=None,
                 provide_label=None, arg_params=None, aux_params=None, dtype=torch.float32):
        self._symbol = symbol
        self._data_names = list(data_names)
        ctx = context[0] if isinstance(context, (list, tuple)) else context
        self._exec = symbol.bind(arg_params, aux_params, device=ctx, dtype=dtype)
        self.output_names = symbol.list_outputs()

    def predict(self, data_batch):
        """-> [ {output_name: tensor} per device ]  (tester.py:38-41)"""
        outs = []
        for idata in data_batch.data:
            out = self._exec.forward(**dict(zip(self._data_names, idata)))
            outs.append(out)
        return outs


def im_detect_device(predictor, data_batch, data_names, scales, cfg, post=None):
    """Forward + fused post-processing, everything left on the device.
    -> (dets (ncls,R,5) f64, counts (ncls,) i32, keep_idx (ncls,R) i32, feat or None, output dict)"""
    output = predictor.predict(data_batch)[0]
    data = dict(zip(data_names, data_batch.data[0]))
    im_shape = data['data'].shape
    rois = output['rois_output']
    scores = output['cls_prob_reshape_output'][0]
    deltas = output['bbox_pred_reshape_output'][0]
    dets, counts, keep_idx = hip.det_postprocess(rois, deltas, scores, im_shape[-2], im_shape[-1], float(scales[0]),
                                                 score_thresh=post['thresh'] if post else 1e-4,
                                                 nms_thresh=cfg.TEST.NMS, max_per_image=cfg.TEST.max_per_image,
                                                 class_agnostic=cfg.CLASS_AGNOSTIC,
                                                 out=post.get('out') if post else None)
    return dets, counts, keep_idx, output.get('choose_feat_output'), output


def im_detect(predictor, data_batch, data_names, scales, cfg):
    """Reference signature and return types (tester.py:130-160): numpy scores, numpy float64
    boxes (decoded, clipped, /scale), the data dict, and the device-resident feature."""
    output_all = predictor.predict(data_batch)
    data_dict_all = [dict(zip(data_names, data_batch.data[i])) for i in range(len(data_batch.data))]
    scores_all, pred_boxes_all = [], []
    for output, data_dict, scale in zip(output_all, data_dict_all, scales):
        rois = output['rois_output']
        im_shape = data_dict['data'].shape
        scores = output['cls_prob_reshape_output'][0]
        bbox_deltas = output['bbox_pred_reshape_output'][0]
        pred_boxes = hip.bbox_pred_clip(rois, bbox_deltas, im_shape[-2], im_shape[-1], float(scale))
        scores_all.append(scores.cpu().numpy())
        pred_boxes_all.append(pred_boxes.cpu().numpy())
    feat = output_all[0].get('choose_feat_output')
    return scores_all, pred_boxes_all, data_dict_all, feat


def im_batch_detect(predictor, data_batch, data_names, scales, cfg):
    """dff_rfcn/core/tester.py:163-189 for the batch symbol: rois of image i are rows
    [i*post_n, (i+1)*post_n) of rois_output (MultiProposal's layout), decoded per image against its
    own im_info and scale.  Returns host arrays like the reference."""
    output_all = predictor.predict(data_batch)
    data_dict_all = [dict(zip(data_names, data_batch.data[i])) for i in range(len(data_batch.data))]
    scores_all, pred_boxes_all = [], []
    for output, data_dict, scale in zip(output_all, data_dict_all, scales):
        im_infos = data_dict['im_info'].cpu().numpy()
        scores = output['cls_prob_reshape_output'][0]
        bbox_deltas = output['bbox_pred_reshape_output'][0]
        rois = output['rois_output']
        post_n = rois.shape[0] // im_infos.shape[0]
        for im_idx in range(im_infos.shape[0]):
            sl = slice(im_idx * post_n, (im_idx + 1) * post_n)
            im_shape = im_infos[im_idx, :2].astype(np.int64)
            pred_boxes = hip.bbox_pred_clip(rois[sl], bbox_deltas[sl], float(im_shape[0]), float(im_shape[1]),
                                            float(scale[im_idx]))
            scores_all.append(scores[sl].cpu().numpy())
            pred_boxes_all.append(pred_boxes.cpu().numpy())
    return scores_all, pred_boxes_all, data_dict_all


class HostRing(object):
    """Detections leave the device through a bounded ring of pinned host buffers: frame i's
    (dets, counts) are copied asynchronously on the stream that produced them into slot i % n, an
    event marks the copy, and a slot is drained into `all_boxes` (compact per-class numpy arrays, what
    the reference keeps, tester.py:272) before it is reused.  Device memory held per frame: none;
    host memory: n slots, independent of the dataset size."""

    def __init__(self, all_boxes, num_classes, R, slots=32):
        self.all_boxes, self.num_classes = all_boxes, num_classes
        # dets and counts back to back in one pinned allocation per slot, as the device side keeps them
        # (core/graphs.py _alloc_post): a frame's results then cross the bus as ONE copy
        n_d, n_c = num_classes * R * 5, (num_classes * 4 + 7) // 8
        self.slots = []
        for _ in range(slots):
            flat = torch.empty(n_d + n_c, dtype=torch.float64).pin_memory()
            self.slots.append(dict(flat=flat, dets=flat[:n_d].view(num_classes, R, 5),
                                   counts=flat[n_d:].view(torch.int32)[:num_classes], event=torch.cuda.Event(), idx=-1))
        self.n = 0

    def _drain(self, slot):
        if slot['idx'] < 0:
            return
        slot['event'].synchronize()
        dets, counts = slot['dets'].numpy(), slot['counts'].numpy()
        for j in range(1, self.num_classes):
            self.all_boxes[j][slot['idx']] = dets[j, :counts[j]].copy()
        slot['idx'] = -1

    def push(self, idx, dets, counts):
        """Queue the copies on the CURRENT stream (the one the frame's post-processing ran on)."""
        slot = self.slots[self.n % len(self.slots)]
        self.n += 1
        self._drain(slot)
        flat = getattr(dets, 'lsfa_flat', None)
        if flat is not None and flat.numel() == slot['flat'].numel():
            slot['flat'].copy_(flat, non_blocking=True)
        else:
            slot['dets'].copy_(dets, non_blocking=True)
            slot['counts'].copy_(counts, non_blocking=True)
        slot['event'].record(torch.cuda.current_stream(dets.device))
        slot['idx'] = idx

    def finish(self):
        for slot in self.slots:
            self._drain(slot)


def pred_eval(gpu_id, key_predictor, cur_predictor, test_data, imdb, cfg, vis=False, thresh=1e-4, logger=None,
              ignore_cache=True):
    """Frame loop of tester.py:192-299.  Returns (all_boxes, frame_ids) with
    all_boxes[cls][image] = (n, 5) float64 array [x1, y1, x2, y2, score]."""
    num_classes = imdb.num_classes if imdb is not None else cfg.dataset.NUM_CLASSES
    data_names = [k[0] for k in test_data.provide_data[0]]
    num_images = test_data.size
    roidb_frame_ids = [x['frame_id'] for x in test_data.roidb]
    all_boxes = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    frame_ids = np.zeros(num_images, dtype=np.int64)
    roidb_idx, roidb_offset, idx = -1, -1, 0
    data_time = net_time = post_time = 0.0
    ring = post_out = None
    feat = None
    t = time.time()
    for im_info, key_frame_flag, data_batch in test_data:
        if ring is None:
            dev = data_batch.data[0][0].device
            R = cfg.TEST.RPN_POST_NMS_TOP_N
            ring = HostRing(all_boxes, num_classes, R)
            # one set of device output buffers for every frame: the copy to the host ring is queued on the
            # same stream right after the post-processing, so the next frame may overwrite them
            from lsfa_amd.core.graphs import _alloc_post
            post_out = _alloc_post(1, num_classes, R, dev)[1]     # dets + counts in one allocation: one copy per frame
        t1 = time.time() - t
        t = time.time()
        scales = [iim_info[0, 2] for iim_info in im_info]
        if key_frame_flag != 2:
            if key_frame_flag == 0:
                feat_old = torch.zeros((1, cfg.network.DFF_FEAT_DIM, 1, 1), device=data_batch.data[0][0].device)
            else:
                feat_old = feat
            data_batch.data[0][-2] = feat_old
            data_batch.provide_data[0][-2] = ('feat_key_old', tuple(feat_old.shape))
            dets, counts, _, feat, _ = im_detect_device(key_predictor, data_batch, data_names, scales, cfg,
                                                        post={'thresh': thresh, 'out': post_out})
        else:
            data_batch.data[0][-1] = feat
            data_batch.provide_data[0][-1] = ('feat_key', tuple(feat.shape))
            dets, counts, _, _, _ = im_detect_device(cur_predictor, data_batch, data_names, scales, cfg,
                                                     post={'thresh': thresh, 'out': post_out})
        if key_frame_flag == 0:
            roidb_idx += 1
            roidb_offset = 0
        else:
            roidb_offset += 1
        frame_ids[idx] = roidb_frame_ids[roidb_idx] + roidb_offset
        ring.push(idx, dets, counts)            # async copy to pinned host memory; no per-frame sync
        t2 = time.time() - t
        t = time.time()
        idx += test_data.batch_size
        data_time += t1
        net_time += t2
        if logger and idx % 50 == 0:
            logger.info('testing {}/{} data {:.4f}s net {:.4f}s'.format(idx, num_images, data_time / idx, net_time / idx))
    t = time.time()
    if ring is not None:
        ring.finish()
    post_time = time.time() - t
    if logger:
        logger.info('done {} frames: data {:.4f}s net {:.4f}s post {:.4f}s per frame'.format(
            num_images, data_time / max(idx, 1), net_time / max(idx, 1), post_time / max(idx, 1)))
    return all_boxes, frame_ids


def pred_eval_pipelined(gpu_id, key_predictor, cur_predictor, test_data, imdb, cfg, thresh=1e-4, logger=None, lanes=2,
                        use_graphs=True, max_pipelines=4, segment=0, key_group=1):
    """pred_eval with the frames of each video pipelined over HIP streams (core/graphs.py
    FramePipeline): same loader, same flags, same launch sequences per frame, same return value.
    One pipeline (captured graphs + static buffers) is built per distinct (height, width, scale) and
    reused by every video of that shape.
    segment / key_group (default off: then every frame's arithmetic is the serial loop's, bit for bit): the batched passes of
    FramePipeline - `segment` non-key frames per pass (KEY_FRAME_INTERVAL - 1 batches whole segments; a shorter run before a video's last
    frame goes frame by frame) and the image-only half of up to `key_group` key frames per pass, for which the loader is asked for the
    coming key frames' images (TestLoader.upcoming_key_frames).  Same detections up to the rounding of a different K cut."""
    from lsfa_amd.core.graphs import FramePipeline
    num_classes = imdb.num_classes if imdb is not None else cfg.dataset.NUM_CLASSES
    data_names = [k[0] for k in test_data.provide_data[0]]
    num_images = test_data.size
    roidb_frame_ids = [x['frame_id'] for x in test_data.roidb]
    all_boxes = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    frame_ids = np.zeros(num_images, dtype=np.int64)
    pipelines = {}
    ring = HostRing(all_boxes, num_classes, cfg.TEST.RPN_POST_NMS_TOP_N)
    roidb_idx, roidb_offset, idx = -1, -1, 0
    fp = None
    # diagnostics (tools/diag_multirank.py): $LSFA_TAP_SUMS=<file> keeps, per frame, (sum |x|, sum x) in float64 of every stage
    # output, computed on the frame's own stream (no synchronisation inside the loop), and writes them as JSON at the end
    tap_file = os.environ.get('LSFA_TAP_SUMS')
    tap_sums = []
    t0 = time.time()
    for im_info, key_frame_flag, data_batch in test_data:
        d = dict(zip(data_names, data_batch.data[0]))
        data = d['data']

        def deliver(bufs, i=idx, flag=key_frame_flag):
            ring.push(i, bufs[0], bufs[1])      # copies queued on the frame's own stream, into pinned host memory
            if tap_file:
                lane = fp.delivering
                if flag == 0:
                    taps, out = dict(fp.first_taps), fp.first_out
                else:
                    taps = dict(lane.taps if hasattr(lane, 'front') else lane.cur_taps)
                    out = lane.out if hasattr(lane, 'front') else lane.cur_out
                taps.update({k: v for k, v in (out or {}).items() if isinstance(v, torch.Tensor)})
                names = sorted(k for k, v in taps.items() if isinstance(v, torch.Tensor) and v.is_floating_point())
                vals = torch.stack([torch.stack([taps[k].double().abs().sum(), taps[k].double().sum()]) for k in names])
                tap_sums.append((i, flag, names, vals))

        if key_frame_flag == 0:
            shape_key = (int(data.shape[-2]), int(data.shape[-1]), float(im_info[0][0, 2]))
            if fp is not None:
                fp.join()
            fp = pipelines.get(shape_key)
            if fp is not None:
                pipelines[shape_key] = pipelines.pop(shape_key)          # most recently used last
            if fp is None:
                while len(pipelines) >= max_pipelines:                   # a dataset with many frame shapes: evict the oldest
                    pipelines.pop(next(iter(pipelines))).close()
                fp = pipelines[shape_key] = FramePipeline(key_predictor._exec, cur_predictor._exec, cfg, shape_key[0],
                                                          shape_key[1], data.device, thresh=thresh,
                                                          use_graphs=use_graphs, lanes=max(lanes, 2) if segment else lanes, taps=bool(tap_file),
                                                          segment=segment, key_group=key_group)
                fp.set_scale(shape_key[2])
                if logger:
                    logger.info('pipeline %dx%d streams: %s' % (shape_key[0], shape_key[1], fp.layout_used))
            deliver(fp.first_frame(data))
            if not fp.captured:
                fp.capture()
            roidb_idx += 1
            roidb_offset = 0
        else:
            upcoming = test_data.upcoming_key_frames(key_group - 1) if (key_frame_flag == 1 and key_group > 1) else None
            ready = torch.cuda.Event()
            ready.record()          # the loader produced this frame's (and the announced key frames') tensors on the current stream
            if key_frame_flag == 1:
                fp.key_frame(data, deliver=deliver, ready=ready, upcoming=upcoming)
            else:
                fp.cur_frame(data, d['motion_vector'], d['res_diff'], deliver=deliver, ready=ready)
            roidb_offset += 1
        frame_ids[idx] = roidb_frame_ids[roidb_idx] + roidb_offset
        idx += test_data.batch_size
        if logger and idx % 50 == 0:
            logger.info('queued {}/{} frames, {:.4f}s per frame'.format(idx, num_images, (time.time() - t0) / idx))
    if fp is not None:
        fp.join()
    ring.finish()
    torch.cuda.synchronize()
    # an overflow of an fp16 scale anywhere in the run is an error, not a detection (lsfa_status_check)
    key_predictor._exec.check_status()
    cur_predictor._exec.check_status()
    if tap_file:
        import json
        with open(tap_file + ('.rank%d' % gpu_id if gpu_id else ''), 'w') as f:
            json.dump([{'frame': int(frame_ids[i]), 'flag': int(flag), 'sums': {n: [float(a), float(b)] for n, (a, b) in zip(names, vals.cpu().tolist())}}
                       for i, flag, names, vals in tap_sums], f)
    for p_ in pipelines.values():
        p_.close()
    net_time = time.time() - t0
    if logger:
        logger.info('done {} frames: {:.4f}s per frame ({:.1f} frames/s)'.format(num_images, net_time / max(idx, 1),
                                                                                 idx / max(net_time, 1e-9)))
    return all_boxes, frame_ids
