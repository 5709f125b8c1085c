"""Training-side pieces that share the detector's tensors: forward evaluation of the
CenterNet detection loss and the target encoder, on the GPU through the C ABI.

    to_output_map(boxes, lms, c, s, out_w, out_h, ...)      dataset/dataset.py:146,160-179: flip + affine into the output map
    encode_targets(boxes, lms, counts, h, w, max_objs)      dataset/dataset.py:160-217 + utils/image.py:95-141
    ctdet_loss(heads, batch, ...)                           model/losses.py:347-374 on explicit head maps
    Engine-level: ctdet_loss_last_forward(engine, batch)    the same on the head maps of the last forward

`batch` is the dict the reference's Dataset returns (dataset.py:223-226): hm, reg_mask, ind, wh, reg (the
reference calls the offset target 'reg'), lm_mask, lm_ind, lm (= 'landmarks').  No CPU path: every function
raises when the HIP library is missing."""
import ctypes as C

import numpy as np

from . import _lib

_KEYS = ("hm", "reg_mask", "ind", "wh", "reg", "lm_mask", "lm_ind", "lm")


def _targets(batch, B):
    t = {"hm": np.ascontiguousarray(batch["hm"], np.float32).reshape(B, -1),
         "reg_mask": np.ascontiguousarray(batch["reg_mask"], np.uint8), "ind": np.ascontiguousarray(batch["ind"], np.int64),
         "wh": np.ascontiguousarray(batch["wh"], np.float32), "reg": np.ascontiguousarray(batch["reg"], np.float32),
         "lm_mask": np.ascontiguousarray(batch["lm_mask"], np.uint8), "lm_ind": np.ascontiguousarray(batch["lm_ind"], np.int64),
         "lm": np.ascontiguousarray(batch["lm"], np.float32)}
    M = t["ind"].shape[1]
    if t["ind"].shape != (B, M) or t["wh"].shape != (B, M, 2) or t["reg"].shape != (B, M, 2) or t["lm"].shape != (B, M, 10):
        raise ValueError("target shapes must be [B,M], [B,M,2], [B,M,2], [B,M,10]")
    return t, M


def to_output_map(boxes, lms, center, scale, output_w, output_h, rot=0, flipped=False, width=None):
    """The per-object coordinate step in front of the target encoder (dataset/dataset.py:146,160-179): optional
    horizontal flip (:164-172, incl. the left/right landmark swap), then ``affine_transform`` of both box corners and
    the five landmark points with ``get_affine_transform(c, s, rot, [output_w, output_h])``.  boxes [n,4], lms [n,10]
    (lms[k][0] < 0 = no landmarks) in source-image pixels -> float32 copies in output-map coordinates, ready for
    ``encode_targets``.  Host numpy (dataset preparation, as in the reference)."""
    from .post_process import get_affine_transform, affine_transform
    t = get_affine_transform(center, scale, rot, [output_w, output_h])
    boxes = np.array(boxes, np.float32).reshape(-1, 4).copy()
    lms = np.array(lms, np.float32).reshape(-1, 10).copy()
    for k in range(len(boxes)):
        bbox, lm = boxes[k], lms[k]
        if flipped:
            bbox[[0, 2]] = width - bbox[[2, 0]] - 1
            if lm[0] >= 0:
                lm[0::2] = width - lm[0::2] - 1
                tmp = lm.copy()
                lm[0:2], lm[2:4], lm[6:8], lm[8:10] = tmp[2:4], tmp[0:2], tmp[8:10], tmp[6:8]
        bbox[:2] = affine_transform(bbox[:2], t)
        bbox[2:] = affine_transform(bbox[2:], t)
        if lm[0] >= 0:
            for j in range(5):
                lm[2 * j:2 * j + 2] = affine_transform(lm[2 * j:2 * j + 2], t)
    return boxes, lms


def encode_targets(boxes, lms, counts, h, w, device=0):
    """boxes [B,M,4] (x1,y1,x2,y2) and lms [B,M,10] in OUTPUT-MAP coordinates, counts [B] -> dict of target arrays."""
    boxes = np.ascontiguousarray(boxes, np.float32)
    lms = np.ascontiguousarray(lms, np.float32)
    counts = np.ascontiguousarray(counts, np.int32)
    B, M = boxes.shape[:2]
    out = {"hm": np.zeros((B, 1, h, w), np.float32), "wh": np.zeros((B, M, 2), np.float32), "reg": np.zeros((B, M, 2), np.float32),
           "ind": np.zeros((B, M), np.int64), "reg_mask": np.zeros((B, M), np.uint8), "landmarks": np.zeros((B, M, 10), np.float32),
           "lm_ind": np.zeros((B, M), np.int64), "lm_mask": np.zeros((B, M), np.uint8)}
    _lib.check(_lib.lib().cf_op_encode_targets(device, _lib.ptr(boxes), _lib.ptr(lms), _lib.ptr(counts), B, h, w, M,
                                               _lib.ptr(out["hm"]), _lib.ptr(out["wh"]), _lib.ptr(out["reg"]), _lib.ptr(out["ind"]),
                                               _lib.ptr(out["reg_mask"]), _lib.ptr(out["landmarks"]), _lib.ptr(out["lm_ind"]),
                                               _lib.ptr(out["lm_mask"])), op=True)
    return out


def ctdet_loss(heads, batch, hm_w=1.0, wh_w=0.1, off_w=1.0, lm_w=1.0, device=0):
    """heads: dict of RAW NCHW maps hm [B,1,h,w] (logits), wh, reg, lm.  Returns float32 [loss, hm, wh, off, lm]."""
    hm = np.ascontiguousarray(heads["hm"], np.float32)
    B, _, h, w = hm.shape
    wh, reg, lm = (np.ascontiguousarray(heads[k], np.float32) for k in ("wh", "reg", "lm"))
    t, M = _targets(batch, B)
    wts = np.array([hm_w, wh_w, off_w, lm_w], np.float32)
    out = np.zeros((5,), np.float32)
    _lib.check(_lib.lib().cf_op_ctdet_loss(device, _lib.ptr(hm), _lib.ptr(wh), _lib.ptr(reg), _lib.ptr(lm), B, h, w,
                                           _lib.ptr(t["hm"]), _lib.ptr(t["reg_mask"]), _lib.ptr(t["ind"]), _lib.ptr(t["wh"]),
                                           _lib.ptr(t["reg"]), _lib.ptr(t["lm_mask"]), _lib.ptr(t["lm_ind"]), _lib.ptr(t["lm"]),
                                           M, _lib.ptr(wts), _lib.ptr(out)), op=True)
    return out


def ctdet_loss_last_forward(engine, batch, hm_w=1.0, wh_w=0.1, off_w=1.0, lm_w=1.0):
    """CtdetLoss on the head maps of ``engine``'s last forward (they stay on the GPU)."""
    t, M = _targets(batch, engine.last_B)
    wts = np.array([hm_w, wh_w, off_w, lm_w], np.float32)
    out = np.zeros((5,), np.float32)
    engine._chk(engine._L.cf_ctdet_loss(engine._h, _lib.ptr(t["hm"]), _lib.ptr(t["reg_mask"]), _lib.ptr(t["ind"]), _lib.ptr(t["wh"]),
                                        _lib.ptr(t["reg"]), _lib.ptr(t["lm_mask"]), _lib.ptr(t["lm_ind"]), _lib.ptr(t["lm"]),
                                        M, _lib.ptr(wts), _lib.ptr(out)))
    return out
