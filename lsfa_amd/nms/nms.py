"""lib/nms/nms.py's interface on top of the `_nms` entry point of liblsfa_hip.so.

`py_nms_wrapper(thresh)`, `cpu_nms_wrapper(thresh)`, `gpu_nms_wrapper(thresh, device_id)` return a
callable `dets (n, 5) -> keep` like the reference's (nms.py:19-34); `gpu_nms(dets, thresh,
device_id)` is the Cython module's function (gpu_nms.pyx, recovered in gpu_nms.cu:1488-1806): sort
by score with `scores.argsort()[::-1]`, run `_nms` on the sorted float32 boxes, map the survivors
back through the order.  There is no CPU implementation behind the `py_` / `cpu_` names here — all
three run the HIP kernel (float32 IoUs, as the reference's GPU path); the frame loop itself does
not use them (tester.py's 30 NMS calls per frame are one fused launch, lsfa_det_postprocess).
"""
import numpy as np

from lsfa_amd import hip


def gpu_nms(dets, thresh, device_id=0):
    dets = np.asarray(dets)
    if dets.shape[0] == 0:
        return []
    order = dets[:, 4].argsort()[::-1]
    sorted_dets = np.ascontiguousarray(dets[order, :5], dtype=np.float32)
    keep = hip.nms_host(sorted_dets, float(thresh), int(device_id))
    return list(order[keep])


def nms(dets, thresh):
    return gpu_nms(dets, thresh, 0)


def py_nms_wrapper(thresh):
    def _nms(dets):
        return nms(dets, thresh)
    return _nms


def cpu_nms_wrapper(thresh):
    def _nms(dets):
        return gpu_nms(dets, thresh, 0)
    return _nms


def gpu_nms_wrapper(thresh, device_id):
    def _nms(dets):
        return gpu_nms(dets, thresh, device_id)
    return _nms
