#!/usr/bin/env python3
"""Host-fed steps: a page-locked uint8 batch (64 x 640 x 640 x 3 = 78.6 MB) copied to the device every step, one context and the ring of two;
(Measured once with the copy split into two halves on the device's two copy streams: 53.3 against 53.4 GB/s -- the link is the limit,
not the engine; not kept.)"""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa
B, S, K = 64, 640, 100
rng = np.random.default_rng(0)
out = {}
for depth in (1, 2):
    ring = cfa.EngineRing(S, S, depth=depth, max_batch=B, dtype="bf16")
    hosts = []
    for j in range(2 * depth):
        a = ring.engines[0].pinned_array((B, S, S, 3)); a[...] = rng.integers(0, 256, a.shape, dtype=np.uint8); hosts.append(a)
    outs = [(e.device_alloc(B * K * 24), e.device_alloc(B * K * 40), e.device_alloc(B * K * 8)) for e in ring.engines]
    def step(i):
        e = ring.engines[i % depth]; o = outs[i % depth]
        e.forward_enqueue(hosts[i % len(hosts)]); e.decode_topk_device(K, o[0], o[1], o[2])
    for i in range(8): step(i)
    ring.synchronize()
    rates = []
    for _ in range(7):
        t0 = time.perf_counter()
        for i in range(20): step(i)
        ring.synchronize()
        rates.append(B * 20 / (time.perf_counter() - t0))
    out["depth_%d" % depth] = round(float(np.median(rates)), 1)
    ring.close()
print(json.dumps({"images_per_s": out, "GBps": {k: round(v * S * S * 3 / 1e9, 1) for k, v in out.items()}}))
