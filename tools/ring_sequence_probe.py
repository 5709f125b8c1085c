#!/usr/bin/env python3
"""Rings created one after the other in ONE process (each closed before the next): rate and pipe map per ring, per placement mode."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa
B, S, K = 64, 640, 100
rng = np.random.default_rng(0)
placement = sys.argv[1]
depths = [int(d) for d in (sys.argv[2] if len(sys.argv) > 2 else "2,2,1,2,3,2").split(",")]
res = []
for depth in depths:
    ring = cfa.EngineRing(S, S, depth=depth, max_batch=B, dtype="bf16", placement=placement)
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
    for _ in range(7):
        t0 = time.perf_counter()
        for i in range(20): step(i)
        ring.synchronize()
        rates.append(B * 20 / (time.perf_counter() - t0))
    clash = []
    if depth >= 2:
        for i in range(depth):
            for j in range(i):
                if ring.engines[i].queue_shared(16, ring.engines[j], 0):
                    clash.append("main%d/main%d" % (j, i))
    res.append({"depth": depth, "k_img_s": round(float(np.median(rates)) / 1e3, 2), "main_pipe_clashes": clash, "rerolls": ring.queue_rerolls})
    for p in xs: e0.device_free(p)
    ring.close()
print(json.dumps({"placement": placement, "rings": res}))
