#!/usr/bin/env python3
"""Does splitting the batch over two contexts (two HIP streams, concurrent kernels) hide per-kernel tails?
Throughput of 1 x B=64 vs 2 x B=32 vs 4 x B=16 contexts enqueued round-robin."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa

S, K, TOTAL = 640, 100, 64
rng = np.random.default_rng(0)
for nctx in (1, 2, 4):
    B = TOTAL // nctx
    engs = [cfa.Engine(S, S, max_batch=B, dtype="bf16") for _ in range(nctx)]
    bufs = []
    for e in engs:
        x = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
        d = e.device_alloc(x.nbytes); e.memcpy_h2d(d, x)
        outs = (e.device_alloc(B * K * 6 * 4), e.device_alloc(B * K * 10 * 4), e.device_alloc(B * K * 8))
        bufs.append((d, outs))
    def step():
        for e, (d, o) in zip(engs, bufs):
            e.forward_enqueue(d, on_device=True, B=B, in_format=0)
            e.decode_topk_device(K, *o)
    for _ in range(5): step()
    for e in engs: e.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n): step()
    for e in engs: e.synchronize()
    dt = time.perf_counter() - t0
    print("%d context(s) x B=%d: %.3f ms per %d images -> %.0f img/s" % (nctx, B, dt / n * 1e3, TOTAL, TOTAL * n / dt))
    for e in engs: e.close()
