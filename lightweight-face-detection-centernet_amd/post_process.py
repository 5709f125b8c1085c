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


def _affine_general(center, scale, rot, output_size, shift, inv):
    """utils/image.py:27-60 in full (rotation, shift): the three point pairs are built in float32 exactly as the
    reference's numpy code does, the 2x3 map is solved in float64 where the reference calls cv2.getAffineTransform."""
    scale = np.asarray(scale, np.float32)
    src_w, (dst_w, dst_h) = scale[0], output_size
    rot_rad = np.pi * rot / 180
    sn, cs = np.sin(rot_rad), np.cos(rot_rad)
    sp = [0, src_w * -0.5]
    src_dir = [sp[0] * cs - sp[1] * sn, sp[0] * sn + sp[1] * cs]                 # get_dir, :74-81
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src, dst = np.zeros((3, 2), np.float32), np.zeros((3, 2), np.float32)
    src[0, :] = center + scale * shift
    src[1, :] = center + src_dir + scale * shift
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    for pts in (src, dst):                                                        # get_3rd_point, :69-71
        d = pts[0] - pts[1]
        pts[2] = pts[1] + np.array([-d[1], d[0]], np.float32)
    a, b = (dst, src) if inv else (src, dst)
    A = np.concatenate([a.astype(np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(A, b.astype(np.float64)).T


def get_affine_transform(center, scale, rot, output_size, shift=None, inv=0):
    """Same signature as utils/image.py:27-60; returns the 2x3 float64 matrix.  rot == 0, shift == 0, inv == 1 (all the
    detection path uses, utils/post_process.py:83-90) comes from the library -- the very matrix the decode kernel
    applies; rotation / shift / the forward direction (dataset side, dataset/dataset.py:146) are host float64."""
    if not isinstance(scale, (np.ndarray, list, tuple)):
        scale = np.array([scale, scale], dtype=np.float32)
    shift0 = shift is None or not np.any(np.asarray(shift) != 0)
    if rot == 0 and shift0 and inv:
        t = np.empty(6, np.float64)
        _lib.check(_lib.lib().cf_affine_from_center_scale(float(center[0]), float(center[1]), float(scale[0]),
                                                          int(output_size[0]), int(output_size[1]), _lib.ptr(t)))
        return t.reshape(2, 3)
    shift = np.zeros(2, np.float32) if shift is None else np.asarray(shift, np.float32)
    return _affine_general(np.asarray(center, np.float32), scale, rot, output_size, shift, inv)


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
