"""CenterNet post-process on the GPU path: map decoded boxes from heat-map coordinates back to
source-image coordinates (the reference's ``utils/post_process.py:83-100`` ``ctdet_post_process``
with ``utils/image.py:19-66`` ``transform_preds`` / ``get_affine_transform`` / ``affine_transform``).

The reference computes the 2x3 matrix with ``cv2.getAffineTransform`` and applies it per point in
Python loops; here the matrix comes from the library (same float32 point construction, float64
3-point solve) and the per-point products run in a HIP kernel -- fused into the top-K decode epilogue
when used through ``Engine.decode_topk(post=(c, s))``.  cv2 is not installed where this was built:
parity at ``cv2.getAffineTransform`` itself is pinned analytically (tests), not against cv2.
"""
import ctypes as C

import numpy as np

from . import _lib


def get_affine_transform(center, scale, rot, output_size, shift=None, inv=0):
    """Same signature as utils/image.py:27-60.  Only rot == 0, shift == 0 (all the detection path
    uses) is supported; returns the 2x3 float64 matrix."""
    if rot != 0 or (shift is not None and np.any(np.asarray(shift) != 0)):
        raise NotImplementedError("only rot=0, shift=0 is on the detection path")
    if not isinstance(scale, (np.ndarray, list, tuple)):
        scale = np.array([scale, scale], dtype=np.float32)
    t = np.empty(6, np.float64)
    _lib.check(_lib.lib().cf_affine_from_center_scale(float(center[0]), float(center[1]), float(scale[0]),
                                                      int(output_size[0]), int(output_size[1]), _lib.ptr(t)))
    m = t.reshape(2, 3)
    if inv:
        return m
    full = np.vstack([m, [0.0, 0.0, 1.0]])
    return np.linalg.inv(full)[:2]


def affine_transform(pt, t):
    """utils/image.py:63-66."""
    new_pt = np.array([pt[0], pt[1], 1.0], dtype=np.float32).T
    return np.dot(t, new_pt)[:2]


def transform_preds(coords, center, scale, output_size, device=0):
    """utils/image.py:19-24: coords [N,2] heat-map coordinates -> [N,2] float64 source coordinates."""
    coords = np.asarray(coords)
    d = np.zeros((1, coords.shape[0], 4), np.float32)
    d[0, :, 0:2] = coords[:, 0:2]
    s = np.asarray(scale, np.float32).reshape(-1)
    s2 = np.array([[s[0], s[-1]]], np.float32)
    c2 = np.asarray(center, np.float32).reshape(1, 2)
    _lib.check(_lib.lib().cf_op_ctdet_post_process(device, _lib.ptr(d), _lib.ptr(c2), _lib.ptr(s2), 1, coords.shape[0], 4,
                                                   int(output_size[0]), int(output_size[1])), op=True)
    return d[0, :, 0:2].astype(np.float64)


def ctdet_post_process(dets, c, s, h, w, num_classes, device=0):
    """utils/post_process.py:83-100.  dets [B,K,>=6] (x1,y1,x2,y2,score,...,cls) in heat-map units,
    modified in place like the reference; returns the list of {class_id(1-based): [[x1,y1,x2,y2,score],...]}."""
    dets = np.asarray(dets)
    B, K, dim = dets.shape
    work = np.ascontiguousarray(dets, dtype=np.float32)
    c2 = np.ascontiguousarray(np.asarray(c, np.float32).reshape(B, 2))
    s_arr = np.asarray(s, np.float32)
    s2 = np.ascontiguousarray(np.repeat(s_arr.reshape(B, 1), 2, axis=1) if s_arr.size == B else s_arr.reshape(B, 2))
    _lib.check(_lib.lib().cf_op_ctdet_post_process(device, _lib.ptr(work), _lib.ptr(c2), _lib.ptr(s2), B, K, dim,
                                                   int(w), int(h)), op=True)
    dets[...] = work
    ret = []
    for i in range(B):
        top_preds = {}
        classes = dets[i, :, -1]
        for j in range(num_classes):
            inds = (classes == j)
            top_preds[j + 1] = np.concatenate([dets[i, inds, :4].astype(np.float32),
                                               dets[i, inds, 4:5].astype(np.float32)], axis=1).tolist()
        ret.append(top_preds)
    return ret
