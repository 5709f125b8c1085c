#!/usr/bin/env python3
"""Golden fixture for the target encoder driven through the reference's OWN ``CenterFaceData.__getitem__``
(dataset/dataset.py:93-245), not through a restatement of its loop:

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_goldens_getitem.py   ->  tests/golden/train_getitem.npz

The reference class is imported from /root/reference and run in this container with the modules the image lacks stubbed:
``cv2`` (imread -> a blank image of the requested size, warpAffine -> a blank canvas: pixel content does not enter the
targets; getAffineTransform -> a float64 3-point solve, the same statement of it the oracle uses: parity with a cv2 BINARY
stays unpinned) and ``matplotlib``.  split = "train" with numpy's global RNG seeded per sample, so the
random scale / centre / flip of :131-140 are exercised; the draws are recovered by replaying the same four RNG calls and
checked against the (c, s) the reference passed to its own get_affine_transform.
Data only is written: raw annotations, image size, (c, s, flipped) and the dict __getitem__ returned.
"""
import os
import sys
import tempfile
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

SHAPES = {}          # image path -> (h, w)


def _solve_affine(src, dst):
    src = np.asarray(src, np.float64); dst = np.asarray(dst, np.float64)
    a = np.concatenate([src, np.ones((3, 1))], axis=1)
    return np.linalg.solve(a, dst).T                     # 2 x 3, float64 like cv2


cv2 = types.ModuleType("cv2")
cv2.INTER_LINEAR = 1
cv2.imread = lambda path: np.zeros(SHAPES[path] + (3,), np.uint8)
cv2.warpAffine = lambda img, m, size, flags=None: np.zeros((size[1], size[0], 3), np.uint8)
cv2.getAffineTransform = _solve_affine
cv2.COLOR_BGR2GRAY = 6
cv2.cvtColor = lambda img, code: np.asarray(img)[..., 0] * 0.0      # colour augmentation of the (blank) input image: not part of the targets
sys.modules["cv2"] = cv2
for name in ("matplotlib", "matplotlib.pyplot"):
    sys.modules[name] = types.ModuleType(name)
sys.path.insert(0, REF)
import dataset.dataset as D                               # noqa: E402


def main():
    rng = np.random.default_rng(77)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "images"))
    sizes = [(480, 640), (640, 480), (720, 1024), (333, 500), (600, 600), (256, 320)]
    lines, raw = [], []
    for i, (h, w) in enumerate(sizes):
        name = "img%d.jpg" % i
        open(os.path.join(tmp, "images", name), "w").close()
        SHAPES[os.path.join(tmp, "images/") + name] = (h, w)
        lines.append("# " + name)
        n = [9, 14, 30, 3, 1, 6][i]
        anns = np.zeros((n, 20), np.float64)
        for k in range(n):
            bw, bh = rng.uniform(4, 0.45 * w), rng.uniform(4, 0.45 * h)
            x, y = rng.uniform(-10, w - bw + 10), rng.uniform(-10, h - bh + 10)      # some boxes stick out of the image
            anns[k, :4] = np.round([x, y, bw, bh], 2)
            if k % 4 != 3:
                for j in range(5):
                    anns[k, 4 + 3 * j] = round(rng.uniform(x, x + bw), 3)
                    anns[k, 5 + 3 * j] = round(rng.uniform(y, y + bh), 3)
                    anns[k, 6 + 3 * j] = 0.0
            else:
                anns[k, 4:19] = -1.0                        # WIDER's "no landmarks" rows
            anns[k, 19] = 0.5
        for a in anns:
            lines.append(" ".join(repr(float(v)) for v in a))
        raw.append(anns)
    txt = os.path.join(tmp, "label.txt")
    open(txt, "w").write("\n".join(lines) + "\n")
    ds = D.CenterFaceData(txt, split="train", debug=False)
    assert len(ds) == len(sizes)

    seen = []
    orig = D.get_affine_transform

    def spy(c, s, rot, out, *a, **k):
        seen.append((np.array(c, np.float32).copy(), np.float32(s), tuple(out)))
        return orig(c, s, rot, out, *a, **k)
    D.get_affine_transform = spy

    out = {"n_samples": np.int32(len(sizes))}
    for i, (h, w) in enumerate(sizes):
        seed = 1000 + i
        np.random.seed(seed)
        del seen[:]
        ret = ds[i]
        c_ref, s_ref, _ = seen[0]
        # replay the four draws of dataset.py:131-140
        np.random.seed(seed)
        s = max(h, w) * 1.0 * np.random.choice(np.arange(0.6, 1.4, 0.1))
        wb, hb = ds._get_border(128, w), ds._get_border(128, h)
        cx = np.random.randint(low=wb, high=w - wb); cy = np.random.randint(low=hb, high=h - hb)
        flipped = bool(np.random.random() < 0.5)
        c = np.array([cx, cy], np.float32)
        if flipped:
            c[0] = w - c[0] - 1
        assert np.array_equal(c, c_ref) and np.float32(s) == s_ref, (i, c, c_ref, s, s_ref)
        out["s%d_size" % i] = np.array([h, w], np.int32)
        out["s%d_anns" % i] = raw[i]
        out["s%d_c" % i] = c; out["s%d_s" % i] = np.float32(s); out["s%d_flipped" % i] = np.bool_(flipped)
        for k in ("hm", "lm", "reg_mask", "ind", "wh", "reg", "lm_ind", "lm_mask"):
            out["s%d_%s" % (i, k)] = np.asarray(ret[k])
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "train_getitem.npz"), **out)
    print("wrote tests/golden/train_getitem.npz:", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith("s2_")})
    print("objects kept per sample:", [int(out["s%d_reg_mask" % i].sum()) for i in range(len(sizes))],
          "with landmarks:", [int(out["s%d_lm_mask" % i].sum()) for i in range(len(sizes))],
          "flipped:", [bool(out["s%d_flipped" % i]) for i in range(len(sizes))])


if __name__ == "__main__":
    main()
