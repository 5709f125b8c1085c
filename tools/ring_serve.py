#!/usr/bin/env python3
"""Serving loop through EngineRing: submit batch i (resident uint8 input), collect batch i - (depth - 1) to host numpy arrays; images/s incl. the hand-over."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa
B, S, K = 64, 640, 100
rng = np.random.default_rng(0)
D = int(os.environ.get("DEPTH", "2")); LAG = D - 1
ring = cfa.EngineRing(S, S, depth=D, max_batch=B, dtype="bf16")
e0 = ring.engines[0]
xs = []
for j in range(4):
    x = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    p = e0.device_alloc(x.nbytes); e0.memcpy_h2d(p, x); xs.append(p)
def loop(n):
    t = []
    for i in range(n):
        t.append(ring.submit(xs[i % 4], K=K, on_device=True, B=B, in_format=0))
        if i >= LAG:
            ring.collect(t[i - LAG])
    for i in range(max(0, n - LAG), n):
        ring.collect(t[i])
loop(10)
ts = []
for _ in range(7):
    t0 = time.perf_counter(); loop(40); ts.append((time.perf_counter() - t0) / 40)
print(json.dumps({"depth": D, "ring_submit_collect_ms_per_batch": round(float(np.median(ts)) * 1e3, 4), "images_per_s": round(B / float(np.median(ts)), 1)}))
