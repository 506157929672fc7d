"""test_rcnn: build symbols, shard videos over ranks, run pred_eval, gather, evaluate.

Mirror of dff_rfcn/function/test_rcnn.py:50-89 + pred_eval_multiprocess (tester.py:301-312), with
one PROCESS per GPU (rank = GPU) instead of one thread per GPU, and the reference's result-file
merge (imagenet_vid.py:245-268) replaced by one all_gather of the detection rows.
"""
import numpy as np
import torch
import torch.distributed as dist

from lsfa_amd.core import parallel
from lsfa_amd.core.loader import TestLoader
from lsfa_amd.core.tester import Predictor, pred_eval, pred_eval_pipelined
from lsfa_amd.symbols.resnet_v1_101_flownet_rfcn import resnet_v1_101_flownet_rfcn


def get_predictor(sym, sym_instance, cfg, arg_params, aux_params, test_data, ctx, dtype=torch.float32):
    """dff_rfcn/function/test_rcnn.py:28-48: infer + check shapes, then bind."""
    data_shape_dict = dict(test_data.provide_data_single)
    sym_instance.sym = sym
    sym_instance.infer_shape(data_shape_dict)
    sym_instance.check_parameter_shapes(arg_params, aux_params, data_shape_dict, is_train=False)
    data_names = [k[0] for k in test_data.provide_data_single]
    return Predictor(sym, data_names, None, context=ctx, provide_data=test_data.provide_data,
                     provide_label=test_data.provide_label, arg_params=arg_params, aux_params=aux_params, dtype=dtype)


def test_rcnn(cfg, roidb, arg_params, aux_params, device=None, thresh=1e-4, logger=None, dtype=torch.float32,
              pipeline=False, max_pipelines=4, segment=0, key_group=1):
    """Returns (rows, frame_ids_local): `rows` = every rank's detections after the final gather.
    pipeline=True runs the frame loop through pred_eval_pipelined (frames of a video overlapped on HIP
    streams) instead of the reference-shaped serial pred_eval; it keeps the captured pipelines of at most `max_pipelines` frame shapes;
    segment / key_group: its batched passes (pred_eval_pipelined)."""
    rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if device is None:
        device = 'cuda:%d' % torch.cuda.current_device()
    key_sym_instance = resnet_v1_101_flownet_rfcn(cfg)
    cur_sym_instance = resnet_v1_101_flownet_rfcn(cfg)
    key_sym = key_sym_instance.get_key_test_symbol(cfg)
    cur_sym = cur_sym_instance.get_cur_test_symbol(cfg)
    shards = parallel.shard_videos([x['frame_seg_len'] for x in roidb], world)
    my_roidb = [roidb[v] for v in shards[rank]]
    if not my_roidb:
        return parallel.gather_rows(np.zeros((0, 7))), np.zeros(0, np.int64)
    test_data = TestLoader(my_roidb, cfg, batch_size=1, shuffle=False, has_rpn=True, device=device)
    key_predictor = get_predictor(key_sym, key_sym_instance, cfg, arg_params, aux_params, test_data, device, dtype)
    cur_predictor = get_predictor(cur_sym, cur_sym_instance, cfg, arg_params, aux_params, test_data, device, dtype)
    run = pred_eval_pipelined if pipeline else pred_eval
    kw = dict(max_pipelines=max_pipelines, segment=segment, key_group=key_group) if pipeline else {}
    all_boxes, frame_ids = run(rank, key_predictor, cur_predictor, test_data, None, cfg, thresh=thresh, logger=logger, **kw)
    rows = parallel.detections_to_rows(all_boxes, frame_ids)
    return parallel.gather_rows(rows), frame_ids
