#!/usr/bin/env python3
"""Small-batch latency of the detect path (forward + top-K decode to the host), hipGraph replay vs
eager launches.  `python tools/latency.py [--size 640] [--batches 1,2,4,8,16]`"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa


def run(eng, x, B, K, iters):
    fwd = lambda: eng.forward_enqueue(x, on_device=True, B=B, in_format=0)   # u8 HWC BGR, resident in HBM
    fwd(); eng.decode_topk(K)
    fwd(); eng.decode_topk(K)                                 # second sighting captures the graph
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fwd()
        eng.decode_topk(K)                                    # blocking: boxes on the host
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, float(np.percentile(ts, 95)) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--batches", default="1,2,4,8,16")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--topk", type=int, default=100)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    bs = [int(b) for b in a.batches.split(",")]
    rng = np.random.default_rng(0)
    x_host = rng.integers(0, 256, (max(bs), a.size, a.size, 3), dtype=np.uint8)
    print("| B | eager ms (p50 / p95) | hipGraph ms (p50 / p95) | img/s (graph) |")
    print("|---:|---:|---:|---:|")
    engs = {g: cfa.Engine(a.size, a.size, max_batch=max(bs), dtype=a.dtype, graph=g) for g in (False, True)}
    xs = {}
    for g, e in engs.items():
        xs[g] = e.device_alloc(x_host.nbytes)
        e.memcpy_h2d(xs[g], x_host)
    for B in bs:
        r = {g: run(engs[g], xs[g], B, a.topk, a.iters) for g in (False, True)}
        print("| %d | %.3f / %.3f | %.3f / %.3f | %.0f |" % (B, r[False][0], r[False][1], r[True][0], r[True][1], B / r[True][0] * 1e3))
    for e in engs.values():
        e.close()
    # the reference's own call: CenterFace(h, w)(img) -- host uint8 image in, thresholded + NMS'ed boxes out (numpy)
    print()
    print("| CenterFace(h, w)(img), one host image | ms (p50 / p95) |")
    print("|---|---:|")
    for (h, w) in ((a.size, a.size), (478, 720)):
        face = cfa.CenterFace(h, w, dtype=a.dtype)
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for _ in range(3):
            face(img)
        ts = []
        for _ in range(a.iters):
            t0 = time.perf_counter(); face(img); ts.append(time.perf_counter() - t0)
        print("| %dx%d%s | %.3f / %.3f |" % (h, w, "" if (h % 32 == 0 and w % 32 == 0) else " (device resize to %dx%d)" % (face.img_h_new, face.img_w_new),
                                          np.median(ts) * 1e3, np.percentile(ts, 95) * 1e3))
        face.close()


if __name__ == "__main__":
    main()
