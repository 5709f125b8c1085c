"""Batched evaluation-side entry points of the reference's ``eval_widerface.py`` on the GPU path.

``get_detections`` (eval_widerface.py:76-90) is the only batched caller of the model in the
reference: forward a batch, sigmoid/clamp the heat map, then per image ``decode`` (:92-110, the "D2"
decoder: the threshold argument IS honoured, offsets are used -- reg channel 1 on x, channel 0 on
y, plus 0.5 -- no landmarks) and the greedy ``nms`` (:112-152).  Here the forward, the threshold
compaction, the box arithmetic and the NMS all run in HIP kernels; this module only marshals.
``bbox_overlap`` (:48-74) and ``evaluate`` (:172-211) -- the recall / precision of a detection set against annotations, the
"box match" of BASELINE.json's metric -- run their IoU matrix and match counts in ``box_match_kernel`` (csrc/cf_decode.hip,
``cf_op_box_match``); ``box_match`` applies the same metric to two detection sets (bench.py: benchmarked mode vs exact mode).
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


def _concat(rows_list, width):
    off = np.zeros(len(rows_list) + 1, np.int32)
    for i, r in enumerate(rows_list):
        off[i + 1] = off[i] + (0 if r is None else len(r))
    cat = np.zeros((max(int(off[-1]), 1), width), np.float32)
    for i, r in enumerate(rows_list):
        if r is not None and len(r):
            cat[off[i]:off[i + 1]] = np.asarray(r, np.float32)[:, :width]
    return np.ascontiguousarray(cat), off


def bbox_overlap(boxes, query_boxes, device=0):
    """eval_widerface.bbox_overlap (:48-74): float64 [N,K] "+1" IoU of every detection against every annotation (float32
    arithmetic, as numpy computes it for the float32 arrays get_detections returns)."""
    b = np.ascontiguousarray(np.asarray(boxes, np.float32).reshape(len(boxes), -1))
    q = np.ascontiguousarray(np.asarray(query_boxes, np.float32).reshape(len(query_boxes), -1))
    N, K = b.shape[0], q.shape[0]
    out = np.zeros((N, K), np.float64)
    if N == 0 or K == 0:
        return out
    boff, qoff = np.array([0, N], np.int32), np.array([0, K], np.int32)
    _lib.check(_lib.lib().cf_op_box_match(device, 1, _lib.ptr(b), b.shape[1], _lib.ptr(boff), _lib.ptr(q), q.shape[1], _lib.ptr(qoff),
                                          0.5, _lib.ptr(out), None), op=True)
    return out


def match_counts(picked_boxes, annot_boxes, threshold=0.5, device=0):
    """For every image: (detections whose best overlap with an annotation exceeds ``threshold``, annotations whose best overlap
    with a detection does) -- the two counts of evaluate (:195-206), all images in ONE device call.  int32 [n_img, 2]."""
    n = len(picked_boxes)
    counts = np.zeros((n, 2), np.int32)
    if n == 0:
        return counts
    b, boff = _concat(picked_boxes, 4)
    q, qoff = _concat(annot_boxes, 4)
    _lib.check(_lib.lib().cf_op_box_match(device, n, _lib.ptr(b), 4, _lib.ptr(boff), _lib.ptr(q), 4, _lib.ptr(qoff),
                                          float(threshold), None, _lib.ptr(counts)), op=True)
    return counts


def _accumulate(picked_boxes, annots, threshold, device, strip_padding=True, reference_empties=True):
    """One batch of evaluate (:180-208): the reference's three empty cases, then the two ratios per image.
    ``reference_empties=False`` (box_match between two detection sets): an image on which BOTH sets are empty is a perfect
    match (1, 1), and a set that is empty on one side only scores (0, 0) -- the reference's bookkeeping credits recall or
    precision there for reasons that have nothing to do with agreement between two detectors."""
    annots = [np.asarray(a, np.float32).reshape(-1, np.asarray(a).shape[-1] if len(a) else 4) for a in annots]
    if strip_padding:
        annots = [a[a[:, 0] != -1] for a in annots]
    counts = match_counts(picked_boxes, annots, threshold, device)
    recall_iter = precision_iter = 0.0
    for j, boxes in enumerate(picked_boxes):
        na = annots[j].shape[0]
        nb = 0 if boxes is None else len(boxes)
        if not reference_empties:
            if nb == 0 and na == 0:
                recall_iter += 1.0
                precision_iter += 1.0
                continue
            if nb == 0 or na == 0:
                continue
        if boxes is None and na == 0:
            continue
        if nb < 1 and na != 0:
            precision_iter += 1.0
            continue
        if na == 0:
            recall_iter += 1.0
            continue
        recall_iter += int(counts[j, 0]) / na
        precision_iter += int(counts[j, 1]) / nb
    return recall_iter / len(picked_boxes), precision_iter / len(picked_boxes)


def evaluate(val_data, model, threshold=0.5, device=0, detections=None):
    """eval_widerface.evaluate (:172-211), same call shape: ``val_data`` iterates batches ``{'input': ..., 'meta': {'gt_det':
    [per-image [M,4+] arrays, rows with x1 == -1 are padding]}}``, ``model`` a centerface_amd.Engine.  Returns (recall,
    precision) as the reference accumulates them -- including its crossed naming: "recall" = detections matching an annotation
    / annotations, "precision" = annotations matched by a detection / detections.  ``detections``: optional callable
    ``(data, model) -> list`` replacing get_detections (precomputed boxes)."""
    recall = precision = 0.0
    n = 0
    for data in val_data:
        annots = data["meta"]["gt_det"]
        picked = (detections or get_detections)(data, model)
        r, p = _accumulate(picked, annots, threshold, device)
        recall += r
        precision += p
        n += 1
    return recall / n, precision / n


def box_match(test_boxes, ref_boxes, threshold=0.5, device=0):
    """The evaluate metric between two detection sets of the same images (``ref_boxes`` in the role of the annotations):
    {"recall", "precision"} in the reference's sense (previous docstring) on the images where both sets have boxes; both sets
    empty = a perfect match, one empty = (0, 0) (evaluate keeps the reference's own empty cases).  Lists of float32 [n,4+]
    (or [] for no boxes)."""
    r, p = _accumulate(list(test_boxes), list(ref_boxes), threshold, device, strip_padding=False, reference_empties=False)
    return {"recall": r, "precision": p, "iou_threshold": threshold, "images": len(test_boxes)}
