#!/usr/bin/env python3
"""hipGraph replay against eager launches of the same forward (B = 64, 640x640, device-resident input, device-output decode): images/s on one
context and on the ring of two.  Under rocprofv3 every kernel NODE of a replayed graph starts ~5.7 us after its predecessor ends (19 gaps = 110 us
of a 1.35 ms forward on one context) while two eager launches in a stream follow each other within 0.3 us (peak_collect -> topk_select)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa
B, S, K = int(os.environ.get("B", 64)), int(os.environ.get("S", 640)), int(os.environ.get("K", 100))
dtype = os.environ.get("DTYPE", "bf16")
rng = np.random.default_rng(0)
out = {}
for depth in (1, 2, 3):
    for graph in (True, False):
        ring = cfa.EngineRing(S, S, depth=depth, max_batch=B, dtype=dtype, graph=graph)
        e0 = ring.engines[0]
        xs = []
        for j in range(4):
            x = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
            p = e0.device_alloc(x.nbytes); e0.memcpy_h2d(p, x); xs.append(p)
        outs = [(e.device_alloc(B * K * 24), e.device_alloc(B * K * 40), e.device_alloc(B * K * 8)) for e in ring.engines]
        def step(i):
            e = ring.engines[i % depth]; o = outs[i % depth]
            e.forward_enqueue(xs[i % 4], on_device=True, B=B, in_format=0); e.decode_topk_device(K, o[0], o[1], o[2])
        for i in range(12): step(i)
        ring.synchronize()
        rates = []
        for _ in range(9):
            t0 = time.perf_counter()
            for i in range(20): step(i)
            ring.synchronize()
            rates.append(B * 20 / (time.perf_counter() - t0))
        out["depth%d_%s" % (depth, "graph" if graph else "eager")] = round(float(np.median(rates)), 1)
        for p in xs: e0.device_free(p)
        ring.close()
print(json.dumps(out))
