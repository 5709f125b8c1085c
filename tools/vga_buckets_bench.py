#!/usr/bin/env python3
"""BASELINE configs[3]: VGA max-side-640 variable input, batch = 128, 1 GPU, through CenterFaceBuckets (images bucketed
by network shape, device resize + forward + D1 threshold decode + NMS + floor rescale, results in input order).
Prints one JSON line with images/s over the whole call (host arrays in, numpy detections out: PCIe + host
post-processing included -- this is the end-to-end API rate, not the kernel rate)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    shapes = [(480, 640), (640, 480), (640, 640), (448, 640), (640, 416)]
    imgs = [rng.integers(0, 256, shapes[i % 5] + (3,), dtype=np.uint8) for i in range(128)]
    out = {}
    for dtype in ("bf16", "fp32"):
        pool = cfa.CenterFaceBuckets(dtype=dtype, max_batch=32, max_buckets=8)
        pool.detect(imgs)                                                          # contexts + graphs warm
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            res = pool.detect(imgs)
            ts.append(time.perf_counter() - t0)
        out[dtype] = {"images_per_s": round(len(imgs) / float(np.median(ts)), 1), "median_s": round(float(np.median(ts)), 4),
                      "detections": int(sum(len(r[0]) for r in res)), "contexts": pool.created}
        pool.close()
    print(json.dumps({"workload": "BASELINE configs[3]: 128 images, 5 VGA-class shapes, CenterFaceBuckets.detect "
                                  "(host uint8 in -> resize/forward/D1 decode/NMS on the GPU -> numpy boxes out)", "result": out}))


if __name__ == "__main__":
    main()
