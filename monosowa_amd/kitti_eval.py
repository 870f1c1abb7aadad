"""Rotated-box overlaps of the KITTI evaluation on MI355X, with the reference's function names and
numpy-in / numpy-out contracts so that `kitti_eval_python/eval.py` can call them in place of its numba-CUDA kernel:

    rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0)    rotate_iou.py:293-330
    bev_box_overlap(boxes, qboxes, criterion=-1)                          eval.py:192-194
    d3_box_overlap(boxes, qboxes, criterion=-1)                           eval.py:226-230
"""
import ctypes
import os

import numpy as np
import torch

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmonosowa_kitti.so")
SYMBOLS = ("mono_rotate_iou_f32", "mono_box3d_overlap_f32", "mono_extract_dets_f32")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError("HIP extension %s is missing: run `python -m monosowa_amd.build`" % _PATH)
        lib = ctypes.CDLL(_PATH)
        P, I, LL = ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong
        for name in SYMBOLS[:2]:
            fn = getattr(lib, name)
            fn.restype = I
            fn.argtypes = [P, P, P, LL, LL, I, P]
        lib.mono_extract_dets_f32.restype = I
        lib.mono_extract_dets_f32.argtypes = [P] * 6 + [I] * 4 + [P]
        _lib = lib
    return _lib


def _overlap(fn_name, boxes, query, width, criterion, device_id):
    dtype = boxes.dtype
    N, K = boxes.shape[0], query.shape[0]
    if N == 0 or K == 0:
        return np.zeros((N, K), dtype=dtype)
    if not torch.cuda.is_available():
        raise RuntimeError("%s needs the GPU (the reference's kernel is GPU-only as well)" % fn_name)
    dev = torch.device("cuda", device_id)
    b = torch.from_numpy(np.ascontiguousarray(boxes, dtype=np.float32).reshape(N, width)).to(dev)
    q = torch.from_numpy(np.ascontiguousarray(query, dtype=np.float32).reshape(K, width)).to(dev)
    out = torch.empty((N, K), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        code = getattr(load(), fn_name)(b.data_ptr(), q.data_ptr(), out.data_ptr(), N, K, int(criterion),
                                        torch.cuda.current_stream().cuda_stream)
    if code:
        raise RuntimeError("%s failed with code %d" % (fn_name, code))
    return out.cpu().numpy().astype(dtype)


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1, device_id=0):
    """boxes [N,5], query_boxes [K,5] = (cx, cy, w, h, angle) -> overlap matrix [N,K] in boxes.dtype."""
    return _overlap("mono_rotate_iou_f32", boxes, query_boxes, 5, criterion, device_id)


def bev_box_overlap(boxes, qboxes, criterion=-1):
    return rotate_iou_gpu_eval(boxes, qboxes, criterion)


def d3_box_overlap(boxes, qboxes, criterion=-1, device_id=0):
    """Camera-frame 3D boxes [*,7] -> 3D overlap [N,K] (BEV intersection x height overlap, one kernel)."""
    return _overlap("mono_box3d_overlap_f32", boxes, qboxes, 7, criterion, device_id)


def extract_dets_device(outputs, topk=50):
    """decode_helper.py:58-111 in one launch: outputs dict of float32 CUDA tensors -> detections [B, topk, 37]."""
    logits = outputs["pred_logits"].contiguous()
    B, Q, C = logits.shape
    t = lambda k: outputs[k].contiguous()
    out = torch.empty((B, topk, 37), dtype=torch.float32, device=logits.device)
    with torch.cuda.device(logits.device):
        code = load().mono_extract_dets_f32(logits.data_ptr(), t("pred_boxes").data_ptr(), t("pred_angle").data_ptr(),
                                            t("pred_3d_dim").data_ptr(), t("pred_depth").data_ptr(), out.data_ptr(), B, Q, C,
                                            int(topk), torch.cuda.current_stream().cuda_stream)
    if code:
        raise RuntimeError("mono_extract_dets_f32 failed with code %d" % code)
    return out
