#!/usr/bin/env python3
"""configs[3] workload for a rocprofv3 timeline: warm up, then three CenterFaceBuckets.detect calls 50 ms apart (the gaps let
tools/vga_timeline_report.py cut the trace into calls).  VGA_PINNED=0: pageable input (staged path)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa
rng = np.random.default_rng(0)
shapes = [(480, 640), (640, 480), (640, 640), (448, 640), (640, 416)]
imgs = [rng.integers(0, 256, shapes[i % 5] + (3,), dtype=np.uint8) for i in range(128)]
if os.environ.get("VGA_PINNED", "1") == "1":
    pimgs = []
    for im in imgs:
        a = cfa.pinned_empty(im.shape); a[...] = im; pimgs.append(a)
    imgs = pimgs
pool = cfa.CenterFaceBuckets(dtype=os.environ.get("VGA_DTYPE", "bf16"), max_batch=int(os.environ.get("VGA_MAXB", "32")), max_buckets=8)
for _ in range(3):
    pool.detect(imgs)
for _ in range(3):
    time.sleep(0.05)
    t0 = time.perf_counter(); pool.detect(imgs); print("detect ms %.3f" % ((time.perf_counter() - t0) * 1e3))
pool.close()
