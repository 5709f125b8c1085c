"""The evaluator-facing pieces of the reference's ``demo.py``: the WIDER-FACE result-file format
(``demo.py:81-87``) and a batched dump loop.  GUI/webcam/``cv2.imshow`` parts are out of scope."""
import os


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
