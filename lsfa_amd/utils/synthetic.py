"""Seeded synthetic ImageNet-VID-shaped clips: frames, stride-16 motion
vectors and residuals with the shapes and value ranges `get_rpn_testbatch` / `transform_mv_res`
produce (lib/rpn/rpn.py:33-52, lib/utils/image.py:202-263), without ffmpeg or a dataset.

A clip is a smooth random field plus three moving rectangles, translated by a per-clip
global motion; `motion_vector` is what the compressed stream would carry after
accumulation back to the key frame (image.py:53-54: negated, in stride-16 cells).
"""
import numpy as np
import torch


class SyntheticClip(object):
    def __init__(self, clip_id, num_frames, height=600, width=1000, key_frame_interval=10, seed=0):
        self.clip_id, self.num_frames = clip_id, num_frames
        self.height, self.width = height, width
        self.key_frame_interval = key_frame_interval
        rs = np.random.RandomState(seed * 100003 + clip_id * 1000)
        self.motion = rs.uniform(-4, 4, 2)                       # px / frame, (dx, dy)
        self.freq = rs.uniform(0.002, 0.02, (8, 2))
        self.phase = rs.uniform(0, 2 * np.pi, (8, 3))
        self.amp = rs.uniform(10, 30, (8, 3))
        self.rects = [dict(x=rs.uniform(100, width - 300), y=rs.uniform(50, height - 250), w=rs.uniform(80, 260),
                           h=rs.uniform(60, 200), v=rs.uniform(-6, 6, 2), color=rs.uniform(0, 255, 3))
                      for _ in range(3)]
        self.fh, self.fw = int(np.ceil(height / 16.0)), int(np.ceil(width / 16.0))
        self.frame_seg_len = num_frames

    def im_info(self):
        return np.array([[self.height, self.width, 1.0]], dtype=np.float32)

    def frame(self, f, device='cpu'):
        """`data`: float32 RGB (1,3,H,W), PIXEL_MEANS = 0 (config.py:171-176 for resnet-101)."""
        g = torch.Generator(device='cpu').manual_seed(1000 * self.clip_id + f)
        ys = torch.arange(self.height, dtype=torch.float32, device=device).view(-1, 1)
        xs = torch.arange(self.width, dtype=torch.float32, device=device).view(1, -1)
        sx, sy = float(self.motion[0] * f), float(self.motion[1] * f)
        img = torch.full((3, self.height, self.width), 115.0, dtype=torch.float32, device=device)
        for k in range(8):
            arg = (xs - sx) * float(self.freq[k, 0]) + (ys - sy) * float(self.freq[k, 1])
            for c in range(3):
                img[c] += float(self.amp[k, c]) * torch.sin(arg + float(self.phase[k, c]))
        for r in self.rects:
            x0 = int(round(r['x'] + (self.motion[0] + r['v'][0]) * f)) % self.width
            y0 = int(round(r['y'] + (self.motion[1] + r['v'][1]) * f)) % self.height
            x1, y1 = min(x0 + int(r['w']), self.width), min(y0 + int(r['h']), self.height)
            for c in range(3):
                img[c, y0:y1, x0:x1] = float(r['color'][c])
        noise = torch.randn((3, self.height, self.width), generator=g) * 8.0
        img = (img + noise.to(device)).clamp_(0, 255).round_()
        return img.unsqueeze(0).contiguous()

    def frame_u8(self, f):
        """The same frame the way a decoder hands it over: (H, W, 3) uint8, BGR, on the host (cv2.imread's layout, image.py:283).
        frame() is integer-valued in [0, 255], so transform(frame_u8(f), zero means, 1.0) IS frame(f), bit for bit."""
        rgb = self.frame(f)[0]                                   # (3, H, W) float32, integers
        return rgb.flip(0).permute(1, 2, 0).to(torch.uint8).contiguous()

    def motion_vector(self, f, key_f, device='cpu'):
        """(1,2,fh,fw): -(displacement accumulated since the key frame)/16 + N(0,0.05)."""
        g = torch.Generator(device='cpu').manual_seed(7000 + 1000 * self.clip_id + f)
        d = (f - key_f) / 16.0
        mv = torch.empty((1, 2, self.fh, self.fw), dtype=torch.float32)
        mv[:, 0] = -float(self.motion[0]) * d
        mv[:, 1] = -float(self.motion[1]) * d
        mv += 0.05 * torch.randn(mv.shape, generator=g)
        return mv.to(device)

    def res_diff(self, f, device='cpu'):
        g = torch.Generator(device='cpu').manual_seed(9000 + 1000 * self.clip_id + f)
        return (4.0 * torch.randn((1, 3, self.fh, self.fw), generator=g)).to(device)


def synthetic_roidb(num_clips, frames_per_clip, height=600, width=1000, key_frame_interval=10, seed=0):
    """roidb entries shaped like ImageNetVID.gt_roidb()'s (lib/dataset/imagenet_vid.py) as far as
    the test path reads them: frame_seg_len, frame_id, pattern + a clip generator."""
    roidb, fid = [], 0
    for c in range(num_clips):
        clip = SyntheticClip(c, frames_per_clip, height, width, key_frame_interval, seed)
        roidb.append({'clip': clip, 'frame_seg_len': frames_per_clip, 'frame_id': fid,
                      'pattern': 'synthetic/%04d/%%06d' % c, 'height': height, 'width': width})
        fid += frames_per_clip
    return roidb
