"""lib/nms/nms.py's interface on top of liblsfa_hip.so.

`py_nms_wrapper(thresh)`, `cpu_nms_wrapper(thresh)`, `gpu_nms_wrapper(thresh, device_id)` return a
callable `dets (n, 5) -> keep` like the reference's (nms.py:19-34).
  * `gpu_nms(dets, thresh, device_id)` is the Cython module's function (gpu_nms.pyx, recovered in
    gpu_nms.cu:1488-1806): sort by score with `scores.argsort()[::-1]`, run `_nms` on the sorted
    float32 boxes, map the survivors back through the order.
  * `nms(dets, thresh)` is the reference's numpy function (nms.py:37-74), which computes in the dtype
    of `dets` — float64 in pred_eval (tester.py:270-271): float64 dets take the float64 kernel
    (lsfa_nms_sorted_f64: numpy's arithmetic, `ovr <= thresh` survives), float32 dets the float32 one.
    The order comes from the same numpy call the reference makes, so ties break the way its numpy breaks them.
There is no CPU implementation behind the `py_` / `cpu_` names — both run on the GPU; the frame loop
itself does not use them (tester.py's 30 NMS calls per frame are one fused launch, lsfa_det_postprocess).
"""
import numpy as np
import torch

from lsfa_amd import hip


def gpu_nms(dets, thresh, device_id=0):
    dets = np.asarray(dets)
    if dets.shape[0] == 0:
        return []
    order = dets[:, 4].argsort()[::-1]
    sorted_dets = np.ascontiguousarray(dets[order, :5], dtype=np.float32)
    keep = hip.nms_host(sorted_dets, float(thresh), int(device_id))
    return list(order[keep])


def nms(dets, thresh, device_id=0):
    dets = np.asarray(dets)
    if dets.shape[0] == 0:
        return []
    if dets.dtype != np.float64:
        return gpu_nms(dets, thresh, device_id)
    order = dets[:, 4].argsort()[::-1]
    boxes = torch.from_numpy(np.ascontiguousarray(dets[order, :4])).to('cuda:%d' % device_id)
    keep, num = hip.nms_sorted_f64(boxes, float(thresh))
    return list(order[keep[:int(num.item())].cpu().numpy()])


def py_nms_wrapper(thresh):
    def _nms(dets):
        return nms(dets, thresh)
    return _nms


def cpu_nms_wrapper(thresh):
    def _nms(dets):
        return nms(dets, thresh)
    return _nms


def gpu_nms_wrapper(thresh, device_id):
    def _nms(dets):
        return gpu_nms(dets, thresh, device_id)
    return _nms
