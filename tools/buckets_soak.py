#!/usr/bin/env python3
"""Soak of CenterFaceBuckets: 300 detect calls over random subsets of twelve raw shapes with four contexts allowed (constant LRU eviction, stream
placement on every new context), page-locked and pageable inputs mixed; every result against a reference pool that never evicts; device and host
memory must not grow."""
import os, sys, time, json, resource
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa
rng = np.random.default_rng(3)
shapes = [(480, 640), (640, 480), (640, 640), (448, 640), (640, 416), (478, 720), (300, 500), (470, 730), (630, 470), (352, 352), (200, 640), (640, 200)]
imgs = {}
for s in shapes:
    imgs[s] = []
    for k in range(6):
        a = rng.integers(0, 256, s + (3,), dtype=np.uint8)
        if k % 2:
            b = cfa.pinned_empty(a.shape); b[...] = a; a = b
        imgs[s].append(a)
ref = cfa.CenterFaceBuckets(dtype="bf16", max_batch=8, max_buckets=16)
want = {s: ref.detect(imgs[s]) for s in shapes}
pool = cfa.CenterFaceBuckets(dtype="bf16", max_batch=8, max_buckets=4)
def free_mb():
    f, t = torch.cuda.mem_get_info(); return f / 1e6
mem0 = rss0 = None
t0 = time.time()
for it in range(int(os.environ.get("ITERS", "300"))):
    sel = [shapes[i] for i in rng.choice(len(shapes), size=int(rng.integers(1, 7)), replace=False)]
    batch, exp = [], []
    for s in sel:
        ks = rng.choice(6, size=int(rng.integers(1, 7)), replace=False)
        batch += [imgs[s][k] for k in ks]; exp += [want[s][k] for k in ks]
    perm = rng.permutation(len(batch))
    got = pool.detect([batch[i] for i in perm])
    for g, i in zip(got, perm):
        assert np.array_equal(g[0], exp[i][0]) and np.array_equal(g[1], exp[i][1]), (it, i)
    if it == 40:
        mem0, rss0 = free_mb(), resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
print(json.dumps({"calls": it + 1, "seconds": round(time.time() - t0, 1), "contexts_created": pool.created, "device_free_MB_drift": round(free_mb() - mem0, 1),
                  "host_maxrss_MB_growth": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024 - rss0, 1)}))
pool.close(); ref.close()
