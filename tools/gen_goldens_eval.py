#!/usr/bin/env python3
"""Golden vectors for the recall/precision half of SURVEY 8f row N2, from the REFERENCE itself (this container only):

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_goldens_eval.py        -> tests/golden/eval_metrics.npz

* ``bbox_overlap`` (eval_widerface.py:48-74) on seeded float32 detections / annotations: overlapping, disjoint, touching
  (iw or ih exactly 0), nested, sub-pixel and large boxes.
* ``evaluate`` (:172-211) with the module's ``get_detections`` replaced by a function that returns prepared detections
  (the forward + D2 decode in front of it has its own goldens, decode_d2.npz / net.npz): the bookkeeping -- padding rows,
  the three empty cases, per-batch means -- is the reference's own code.

Data only is written (inputs + the reference's outputs).  NumPy here is 2.2: the one mixed float32 / python-float division
of bbox_overlap is a float32 division under NEP 50 (oracle/centerface_oracle.py::bbox_overlap says so).
"""
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import gen_goldens as G                                    # noqa: E402


def boxes_case(rng, n, span, frac):
    x1 = rng.uniform(0, span, n); y1 = rng.uniform(0, span, n)
    w = rng.uniform(1, span / 3, n); h = rng.uniform(1, span / 3, n)
    b = np.stack([x1, y1, x1 + w, y1 + h], 1)
    if not frac:
        b = np.round(b)
    return b.astype(np.float32)


def main():
    G.install_stubs()
    import eval_widerface as ew
    rng = np.random.default_rng(20240928)
    out = {"numpy_version": np.array(np.__version__)}
    # ---- bbox_overlap
    cases = []
    for i, (n, k, span, frac) in enumerate([(7, 5, 64, True), (23, 17, 640, True), (12, 9, 100, False), (1, 1, 32, True), (40, 3, 1280, True)]):
        b = boxes_case(rng, n, span, frac)
        q = boxes_case(rng, k, span, frac)
        if i == 2:      # integer boxes: force touching / just-disjoint pairs (iw == 0, iw == 1) and an identical pair
            q[0] = b[0]
            q[1] = [b[1, 2] + 1, b[1, 1], b[1, 2] + 9, b[1, 3]]          # iw = 0: not counted
            q[2] = [b[2, 2], b[2, 1], b[2, 2] + 9, b[2, 3]]              # iw = 1: one shared pixel column
            q[3] = [b[3, 0] + 1, b[3, 1] + 1, b[3, 2] - 1, b[3, 3] - 1]  # nested
        scores = rng.uniform(0.3, 1.0, (n, 1)).astype(np.float32)
        b5 = np.concatenate([b, scores], 1)
        ov = ew.bbox_overlap(b5, q)
        assert ov.dtype == np.float64 and ov.shape == (n, k)
        out["ov%d_boxes" % i] = b5; out["ov%d_query" % i] = q; out["ov%d_out" % i] = ov
        cases.append(i)
    out["ov_cases"] = np.array(cases)
    # ---- evaluate: 3 batches; images with matches, without detections, without annotations, with neither
    picked_all, annots_all = [], []
    for bi, nimg in enumerate([4, 3, 5]):
        picked, annots = [], []
        for j in range(nimg):
            kind = (bi * 5 + j) % 6
            gt = boxes_case(rng, int(rng.integers(1, 6)), 320, True)
            if kind == 3:
                gt = gt[:0]
            if kind in (1, 4) or (kind == 3 and j % 2 == 0):
                det = []
            else:
                jit = gt + rng.normal(0, 4.0, gt.shape).astype(np.float32) if len(gt) else gt
                extra = boxes_case(rng, int(rng.integers(0, 4)), 320, True)
                d4 = np.concatenate([jit, extra], 0) if len(gt) else boxes_case(rng, 2, 320, True)
                det = np.concatenate([d4, rng.uniform(0.35, 1, (len(d4), 1)).astype(np.float32)], 1).astype(np.float32)
            pad = np.full((8, 4), -1.0, np.float32)                    # gt_det padding rows (x1 == -1), dataset/dataset.py style
            pad[:len(gt)] = gt
            picked.append(det); annots.append(pad)
        picked_all.append(picked); annots_all.append(annots)
    val_data = [{"meta": {"gt_det": a}, "_picked": p} for a, p in zip(annots_all, picked_all)]
    ew.get_detections = lambda data, model, *a, **k: data["_picked"]
    ew.tqdm = lambda it, *a, **k: it
    for thr in (0.5, 0.35):
        r, p = ew.evaluate(val_data, None, threshold=thr)
        out["eval_thr%02d" % int(thr * 100)] = np.array([r, p], np.float64)
    out["eval_batches"] = np.array([len(p) for p in picked_all])
    for bi, (picked, annots) in enumerate(zip(picked_all, annots_all)):
        for j, (d, a) in enumerate(zip(picked, annots)):
            out["eval_b%d_i%d_det" % (bi, j)] = np.asarray(d, np.float32).reshape(-1, 5)
            out["eval_b%d_i%d_gt" % (bi, j)] = a
    np.savez_compressed(os.path.join(G.OUT, "eval_metrics.npz"), **out)
    print("wrote eval_metrics.npz:", {k: v.tolist() for k, v in out.items() if k.startswith("eval_thr")})


if __name__ == "__main__":
    main()
