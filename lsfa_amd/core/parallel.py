"""Clip-parallel multi-GPU inference: one process per GPU, no traffic during inference.

Reference: videos are assigned to GPUs greedily by accumulated frame count
(dff_rfcn/function/test_rcnn.py:69-75), one Python thread per GPU (dff_rfcn/core/tester.py:305-309),
results merged as Python lists in one process (lib/dataset/imagenet_vid.py:245-268).
Here: the same assignment computed identically on every rank, one process per GPU, and ONE
all_gather of the detection rows at the end — RCCL over xGMI on GPUs (backend "nccl"), gloo in the
CPU tests.  Rows are [frame_id, cls, score, x1, y1, x2, y2] in float64.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_videos(seg_lens, world_size):
    """-> list (per rank) of video indices; each video goes to the rank with the fewest frames so far."""
    shards = [[] for _ in range(world_size)]
    acc = np.zeros(world_size, dtype=np.int64)
    for vid, n in enumerate(seg_lens):
        g = int(np.argmin(acc))
        shards[g].append(vid)
        acc[g] += int(n)
    return shards


def detections_to_rows(all_boxes, frame_ids):
    """all_boxes[cls][image] (n,5) + frame_ids -> (rows, 7) float64, the det_<set>_all.txt columns
    (lib/dataset/imagenet_vid.py:260-268)."""
    rows = []
    for im_ind, fid in enumerate(frame_ids):
        for cls_ind in range(1, len(all_boxes)):
            dets = all_boxes[cls_ind][im_ind]
            if len(dets) == 0:
                continue
            r = np.empty((dets.shape[0], 7), np.float64)
            r[:, 0], r[:, 1], r[:, 2], r[:, 3:] = fid, cls_ind, dets[:, 4], dets[:, :4]
            rows.append(r)
    return np.vstack(rows) if rows else np.zeros((0, 7), np.float64)


def gather_rows(rows, device=None):
    """all_gather ragged (n_r, 7) row blocks: first the counts, then one padded tensor.  Returns the
    concatenation in rank order on every rank.  Works on any initialised backend; a world of one still issues both collectives
    (that is how a one-GPU box exercises the RCCL path: tests/test_multirank_gpu.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return rows
    world = dist.get_world_size()
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m = max(max(counts), 1)
    pad = torch.zeros((m, 7), dtype=torch.float64, device=device)
    if rows.shape[0]:
        pad[:rows.shape[0]] = torch.from_numpy(np.ascontiguousarray(rows)).to(device)
    out = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(out, pad)
    return np.vstack([o[:c].cpu().numpy() for o, c in zip(out, counts)])
