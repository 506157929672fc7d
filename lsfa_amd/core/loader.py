"""TestLoader: key-frame scheduling and the per-frame input dict.

Mirror of dff_rfcn/core/loader.py:24-141: flag 0 = first frame of a
video, 1 = key frame (every TEST.KEY_FRAME_INTERVAL frames AND the last frame of every video),
2 = non-key frame; input order `data_name` (:41); placeholders of shape (1, DFF_FEAT_DIM, 1, 1)
for feat_key_old / feat_key (:138-139); `data_key` persists across videos while the first frame of
each video overrides `data_key_old` with itself (:119-122).  Frames come from the roidb
entry's clip generator (lsfa_amd.utils.synthetic) and live on the device.
"""
import numpy as np
import torch


class DataBatch(object):
    """mx.io.DataBatch stand-in: data = [[tensor per data_name]] (one list per device)."""

    def __init__(self, data, label=None, pad=0, index=0, provide_data=None, provide_label=None):
        self.data, self.label, self.pad, self.index = data, label, pad, index
        self.provide_data, self.provide_label = provide_data, provide_label


class TestLoader(object):
    def __init__(self, roidb, config, batch_size=1, shuffle=False, has_rpn=False, device='cuda:0'):
        assert batch_size == 1 and not shuffle
        self.cfg, self.roidb, self.batch_size, self.shuffle, self.has_rpn = config, roidb, batch_size, shuffle, has_rpn
        self.device = device
        self.size = int(np.sum([x['frame_seg_len'] for x in self.roidb]))
        self.index = np.arange(self.size)
        self.data_name = ['data', 'im_info', 'data_key', 'data_key_old', 'motion_vector', 'res_diff', 'feat_key_old',
                          'feat_key']
        self.label_name = None
        self.cur_roidb_index = 0
        self.cur_frameid = 0
        self.data_key = None
        self.data_key_old = None
        self.key_frameid = 0
        self.cur_seg_len = 0
        self.key_frame_flag = -1
        self.cur = 0
        self.data = None
        self.label = []
        self.im_info = None
        self._ahead = {}
        self.reset()
        self.get_batch()

    @property
    def provide_data(self):
        return [[(k, tuple(v.shape)) for k, v in zip(self.data_name, idata)] for idata in self.data]

    @property
    def provide_label(self):
        return [None for _ in range(len(self.data))]

    @property
    def provide_data_single(self):
        return [(k, tuple(v.shape)) for k, v in zip(self.data_name, self.data[0])]

    @property
    def provide_label_single(self):
        return None

    def reset(self):
        self.cur = 0

    def iter_next(self):
        return self.cur < self.size

    def __iter__(self):
        return self

    def __next__(self):
        return self.next()

    def next(self):
        if self.iter_next():
            self.get_batch()
            self.cur += self.batch_size
            self.cur_frameid += 1
            if self.cur_frameid == self.cur_seg_len:
                self.cur_roidb_index += 1
                self.cur_frameid = 0
                self.key_frameid = 0
            elif self.cur_frameid - self.key_frameid == self.cfg.TEST.KEY_FRAME_INTERVAL:
                self.key_frameid = self.cur_frameid
            return self.im_info, self.key_frame_flag, DataBatch(data=self.data, label=self.label, pad=0,
                                                                index=self.cur // self.batch_size,
                                                                provide_data=self.provide_data,
                                                                provide_label=self.provide_label)
        raise StopIteration

    def _frame_inputs(self, entry, f, key_f):
        clip = entry['clip']
        data = self._ahead.pop((self.cur_roidb_index, f), None)      # an image upcoming_key_frames already produced: the SAME tensor
        if data is None:
            data = clip.frame(f, self.device)
        return {'data': data, 'im_info': torch.from_numpy(clip.im_info()).to(self.device),
                'motion_vector': clip.motion_vector(f, key_f, self.device), 'res_diff': clip.res_diff(f, self.device)}

    def upcoming_key_frames(self, n):
        """Call right after a KEY frame was returned: the images of the next `n` key frames of the same video (fewer near its end), for a
        caller that computes the image-only part of several key frames at once (FramePipeline.key_frame(upcoming=...)).  Key frames are
        every KEY_FRAME_INTERVAL-th frame and, by the rule of get_batch above (:106-109), the video's last frame.  The tensors are kept and
        handed out again when the iteration reaches those frames."""
        if self.cur_frameid == 0:             # the frame just returned was its video's last: nothing ahead in this video
            return []
        entry = self.roidb[self.cur_roidb_index]
        K, L = self.cfg.TEST.KEY_FRAME_INTERVAL, entry['frame_seg_len']
        out, prev = [], self.cur_frameid - 1       # the (key) frame just returned; key_frameid may already point past it (interval 1)
        while len(out) < n and prev < L - 1:
            f = min(prev + K, L - 1)          # the next multiple of the interval, or the video's last frame if that comes first
            key = (self.cur_roidb_index, f)
            if key not in self._ahead:
                self._ahead[key] = entry['clip'].frame(f, self.device)
            out.append(self._ahead[key])
            prev = f
        return out

    def get_batch(self):
        cur_roidb = self.roidb[self.cur_roidb_index]
        self.cur_seg_len = cur_roidb['frame_seg_len']
        d = self._frame_inputs(cur_roidb, self.cur_frameid, self.key_frameid)
        if self.key_frameid == self.cur_frameid:       # key frame
            self.data_key_old = self.data_key if self.data_key is not None else d['data']
            self.data_key = d['data']
            if self.key_frameid == 0:
                self.data_key_old = d['data']
                self.key_frame_flag = 0
            else:
                self.key_frame_flag = 1
        elif self.cur_frameid + 1 == self.cur_seg_len:  # the last frame of a video is a key frame
            self.data_key_old = self.data_key if self.data_key is not None else d['data']
            self.data_key = d['data']
            self.key_frame_flag = 1
        else:
            self.key_frame_flag = 2
        dim = self.cfg.network.DFF_FEAT_DIM
        placeholder = torch.zeros((1, dim, 1, 1), device=self.device)
        extend = {'data': d['data'], 'im_info': d['im_info'], 'data_key': self.data_key, 'data_key_old': self.data_key_old,
                  'motion_vector': d['motion_vector'], 'res_diff': d['res_diff'], 'feat_key_old': placeholder,
                  'feat_key': placeholder}
        self.data = [[extend[name] for name in self.data_name]]
        self.im_info = [d['im_info'].cpu().numpy()]
