"""The evaluator-facing pieces of the reference's ``demo.py``: image files in (``cv2.imread`` at demo.py:31,74 ->
PIL here), the WIDER-FACE result-file format (``demo.py:81-87``) and a batched dump loop.  GUI/webcam/``cv2.imshow``
parts are out of scope."""
import os

import numpy as np


def imread(path):
    """``cv2.imread(path)`` (demo.py:31,74): uint8 [H, W, 3] in BGR channel order.  Decoded with PIL (cv2 is not
    a dependency); both sit on libjpeg, bit parity of the DECODER with a cv2 build is unpinned."""
    from PIL import Image
    with Image.open(path) as im:
        rgb = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(rgb[:, :, ::-1])


def detect_file(detector, path, threshold=0.3):
    """demo.py:30-38 (``test_image``): read one image file and run ``CenterFace.__call__`` on it.  ``detector`` must
    have been built for the image's (h, w) -- or be a ``CenterFaceBuckets``, which takes any size."""
    frame = imread(path)
    if hasattr(detector, "detect"):
        return detector.detect([frame], threshold)[0]
    return detector(frame, threshold)


def format_wider_result(rel_name, dets):
    """Text of one result file exactly as demo.py:82-87 writes it: image path, count, then
    ``x y w h score`` per box with w = x2 - x1 + 1, h = y2 - y1 + 1 (:86)."""
    lines = ['{:s}\n'.format(rel_name), '{:d}\n'.format(len(dets))]
    for b in dets:
        x1, y1, x2, y2, s = (float(v) for v in b[:5])
        lines.append('{:.1f} {:.1f} {:.1f} {:.1f} {:.3f}\n'.format(x1, y1, (x2 - x1 + 1), (y2 - y1 + 1), s))
    return ''.join(lines)


def write_wider_result(save_path, im_dir, im_name, dets):
    """demo.py:67-68,81-87: save_path/im_dir/im_name.txt."""
    d = os.path.join(save_path, im_dir)
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, im_name + '.txt')
    with open(path, 'w') as f:
        f.write(format_wider_result('%s/%s.jpg' % (im_dir, im_name), dets))
    return path


def dump_event(detector, images, im_dir, names, save_path):
    """Detect a list of same-sized BGR uint8 images in batches with one ``CenterFace`` instance (the
    reference rebuilds the model per image, demo.py:76) and write one result file per image."""
    results = detector.detect_batch(images, threshold=0.05)
    paths = []
    for name, res in zip(names, results):
        dets = res[0] if isinstance(res, tuple) else res
        paths.append(write_wider_result(save_path, im_dir, name, dets))
    return paths
