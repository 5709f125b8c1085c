"""Batched evaluation-side entry points of the reference's ``eval_widerface.py`` on the GPU path.

``get_detections`` (eval_widerface.py:76-90) is the only batched caller of the model in the
reference: forward a batch, sigmoid/clamp the heat map, then per image ``decode`` (:92-110, the "D2"
decoder: the threshold argument IS honoured, offsets are used -- reg channel 1 on x, channel 0 on
y, plus 0.5 -- no landmarks) and the greedy ``nms`` (:112-152).  Here the forward, the threshold
compaction, the box arithmetic and the NMS all run in HIP kernels; this module only marshals.
The recall/precision code (:154-211) is out of scope (training-loop validation).
"""
import ctypes as C

import numpy as np

from . import _lib


def get_detections(data_batch, model, cuda=True, threshold=0.35, nms_thresh=0.3, max_out=1024, size=(640, 640)):
    """Same call shape as the reference: ``data_batch['input']`` is the float32 [B,3,H,W] batch the
    reference feeds to the model (or a uint8 [B,H,W,3] BGR batch), ``model`` a centerface_amd.Engine.
    Returns a list with one float32 [n,5] array (x1,y1,x2,y2,score) per image, ``[]`` when empty --
    what the reference's ``decode`` returns.  ``size``: the reference clamps boxes to a hard-coded (640, 640) whatever
    the input size (eval_widerface.py:88) -- the default here; pass ``None`` to clamp to the engine's input size."""
    del cuda
    x = data_batch["input"] if isinstance(data_batch, dict) else data_batch
    if hasattr(x, "detach"):
        x = x.detach().cpu().numpy()
    out = []
    for i in range(0, len(x), model.max_batch):
        model.forward_enqueue(x[i:i + model.max_batch])
        for boxes, _ in model.decode_threshold(threshold, nms_thresh, max_out, mode="d2", size=size):
            out.append(boxes if len(boxes) else [])
    return out


def decode(heatmap, scale, offset, landmark, size, threshold=0.1, nms_thresh=0.3, device=0):
    """eval_widerface.decode (:92-110) on explicit per-image maps: heatmap [1,h,w] or [h,w],
    scale [2,h,w], offset [2,h,w]."""
    del landmark
    hm = np.squeeze(np.asarray(heatmap, np.float32))
    h, w = hm.shape
    hm = np.ascontiguousarray(hm.reshape(1, 1, h, w))
    wh = np.ascontiguousarray(np.asarray(scale, np.float32).reshape(1, 2, h, w))
    reg = np.ascontiguousarray(np.asarray(offset, np.float32).reshape(1, 2, h, w))
    cap = max(1, min(h * w, 4096))
    while True:
        dets = np.empty((1, cap, 5), np.float32)
        cnt = np.zeros((1,), np.int32)
        _lib.check(_lib.lib().cf_op_decode_threshold_ex(device, 1, _lib.ptr(hm), _lib.ptr(wh), _lib.ptr(reg), None, 1, h, w,
                                                        int(size[0]), int(size[1]), float(threshold), float(nms_thresh), cap,
                                                        _lib.ptr(dets), None, _lib.ptr(cnt)), op=True)
        if int(cnt[0]) <= cap:
            break
        cap = int(cnt[0])
    n = int(cnt[0])
    return dets[0, :n].copy() if n else []


def nms(boxes, scores, nms_thresh, device=0):
    """eval_widerface.nms (:112-152), identical to CenterFace.nms."""
    L = _lib.lib()
    boxes, scores = _lib.f32(boxes), _lib.f32(scores)
    n = boxes.shape[0]
    keep = np.empty((max(n, 1),), np.int32)
    nk = C.c_int32()
    _lib.check(L.cf_op_nms(device, _lib.ptr(boxes), _lib.ptr(scores), n, float(nms_thresh), _lib.ptr(keep), C.byref(nk)), op=True)
    return [int(k) for k in keep[:nk.value]]
