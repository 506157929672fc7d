"""N>1 path on CPU: the reference's greedy video sharding and the final gather of detection rows
over a world_size-2 gloo group; VID mAP restatement on a hand-built case."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lsfa_amd.core import parallel
from lsfa_amd.dataset import vid_eval as ve
from oracle import np_ref


def test_sharding_matches_reference_rule():
    lens = [144, 30, 90, 12, 60, 60, 7]
    assert parallel.shard_videos(lens, 2) == np_ref.shard_videos(lens, 2)
    assert parallel.shard_videos(lens, 8) == np_ref.shard_videos(lens, 8)
    s = parallel.shard_videos(lens, 3)
    assert sorted(v for r in s for v in r) == list(range(len(lens)))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lens = [5, 3, 4]
    mine = parallel.shard_videos(lens, world)[rank]
    rs = np.random.RandomState(100 + rank)
    all_boxes = [[[] for _ in range(sum(lens[v] for v in mine))] for _ in range(4)]
    frame_ids, im = [], 0
    starts = np.concatenate([[0], np.cumsum(lens)])
    for v in mine:
        for f in range(lens[v]):
            frame_ids.append(int(starts[v] + f))
            for c in range(1, 4):
                n = rs.randint(0, 3)
                all_boxes[c][im] = rs.rand(n, 5) if n else []
            im += 1
    rows = parallel.detections_to_rows(all_boxes, frame_ids)
    merged = parallel.gather_rows(rows)
    q.put((rank, rows, merged))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_rows_world2_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = np.vstack([got[0][1], got[1][1]])
    for _, _, merged in got:
        np.testing.assert_array_equal(merged, want)
    assert set(np.unique(want[:, 0]).astype(int)) <= set(range(12))


def test_vid_eval_hand_built_case():
    gt = [{'img_id': 0, 'bbox': [[10, 10, 109, 109], [200, 200, 219, 219]], 'label': [1, 2]},
          {'img_id': 1, 'bbox': [[50, 50, 149, 149]], 'label': [1]}]
    rows = np.array([
        [0, 1, 0.9, 12, 12, 111, 111],     # TP for class 1 (IoU ~0.92)
        [0, 1, 0.8, 10, 10, 109, 109],     # duplicate of an already matched gt -> FP
        [0, 2, 0.7, 201, 201, 220, 220],   # small box: threshold lowered to wh/((w+10)(h+10)) = 0.444 -> TP
        [1, 1, 0.6, 300, 300, 400, 400],   # FP
        [1, 1, 0.5, 52, 48, 151, 147],     # TP
    ], dtype=np.float64)
    ap = ve.vid_eval(rows, gt, 3)
    # class 1: order 0.9 TP, 0.8 FP, 0.6 FP, 0.5 TP; npos 2 -> recall .5,.5,.5,1  precision 1,.5,.33,.5
    assert abs(ap[0] - (0.5 * 1.0 + 0.5 * 0.5)) < 1e-12
    assert abs(ap[1] - 1.0) < 1e-12
    assert abs(ve.gt_threshold([200, 200, 219, 219]) - 400.0 / 900.0) < 1e-12
    assert ve.format_rows(rows[:1]) == ['0 1 0.9000 12.00 12.00 111.00 111.00']
    assert ve.vid_eval(np.zeros((0, 7)), gt, 3).tolist() == [0.0, 0.0]


def test_vid_eval_matches_loop_form_restatement():
    """The vectorised evaluator vs oracle/np_ref.vid_eval_ref (lib/dataset/imagenet_vid_eval.py restated loop
    by loop) on seeded random frames: jittered copies of the ground truth (hits, duplicates, near-threshold
    overlaps, small boxes whose threshold is relaxed), wrong-class copies, background boxes, frames without
    annotation records, score ties, empty ground truths."""
    from oracle import np_ref
    for seed in range(6):
        rs = np.random.RandomState(seed)
        ncls = 6
        gt, rows = [], []
        for f in range(12):
            k = rs.randint(0, 5)
            xy = rs.uniform(0, 300, (k, 2))
            wh = np.where(rs.rand(k, 1) < 0.3, rs.uniform(4, 20, (k, 2)), rs.uniform(30, 200, (k, 2)))
            boxes = np.round(np.concatenate([xy, xy + wh], 1))
            labels = rs.randint(1, ncls, k)
            if f != 7:                                   # frame 7 has detections but no annotation record
                gt.append({'img_id': f, 'bbox': boxes, 'label': labels})
            for b, l in zip(boxes, labels):
                for _ in range(rs.randint(0, 4)):
                    jit = b + rs.uniform(-1, 1, 4) * rs.choice([1.0, 6.0, 25.0])
                    cls = l if rs.rand() < 0.8 else rs.randint(1, ncls)
                    rows.append([f, cls, np.round(rs.uniform(0.05, 1.0), 1), jit[0], jit[1], jit[2], jit[3]])
            for _ in range(rs.randint(0, 3)):
                xy0 = rs.uniform(0, 400, 2)
                rows.append([f, rs.randint(1, ncls), rs.uniform(0.05, 1.0), xy0[0], xy0[1], xy0[0] + 50, xy0[1] + 40])
        rows = np.array(rows, dtype=np.float64)
        rows = rows[rs.permutation(len(rows))]
        for through_text in (True, False):
            want = np_ref.vid_eval_ref(rows, gt, ncls, through_text=through_text)
            got = ve.vid_eval(rows, gt, ncls, through_text=through_text)
            np.testing.assert_array_equal(got, want)
        assert want.max() > 0


@pytest.mark.parametrize("K,n", [(4, 11), (4, 9), (3, 7), (10, 24), (1, 5)])
def test_loader_announces_the_coming_key_frames(K, n):
    """TestLoader.upcoming_key_frames (the look-ahead pred_eval_pipelined's key groups need), on the CPU: after every key frame (flags 0 / 1
    of dff_rfcn/core/loader.py:92-121's state machine, incl. the rule that a video's last frame is a key frame) it names the next two key
    frames of the SAME video - every KEY_FRAME_INTERVAL-th frame, then the last one - never a frame of the next video, and the iteration
    later hands out those very tensors (same storage), in that order."""
    from lsfa_amd.config.config import lsfa_test_config
    from lsfa_amd.core.loader import TestLoader
    from lsfa_amd.utils.synthetic import synthetic_roidb
    cfg = lsfa_test_config(key_frame_interval=K)
    loader = TestLoader(synthetic_roidb(2, n, 32, 48, K), cfg, device='cpu')
    expected_keys = sorted(set(list(range(0, n, K)) + [n - 1]))
    pending, seen, i = [], [], 0
    for im_info, flag, batch in loader:
        d = dict(zip(loader.data_name, batch.data[0]))
        f = i % n
        if f == 0:
            assert flag == 0 and not pending          # nothing announced across a video boundary
            seen = []
        if flag in (0, 1):
            seen.append(f)
            if pending:
                assert pending.pop(0).data_ptr() == d['data'].data_ptr(), (K, n, f)
            up = loader.upcoming_key_frames(2)
            ahead = [k for k in expected_keys if k > f][:2]
            assert len(up) == len(ahead), (K, n, f, len(up), ahead)
            for t, k in zip(up, ahead):
                assert torch.equal(t, synthetic_roidb(2, n, 32, 48, K)[i // n]['clip'].frame(k, 'cpu')), (K, n, f, k)
            known = {p.data_ptr() for p in pending}
            pending += [t for t in up if t.data_ptr() not in known]
        if f == n - 1:
            assert seen == expected_keys, (K, n, seen)
        i += 1
    assert i == 2 * n and not pending
