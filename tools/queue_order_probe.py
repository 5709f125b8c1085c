#!/usr/bin/env python3
"""Does it matter WHICH hardware queues the ring's streams get?  k dummy streams (torch.cuda.Stream, touched) are created before the
two-context ring, shifting the placement of everything after them; rate of the bench step loop (B = 64, 640x640, device outputs)."""
import os, sys, time, json
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa
B, S, K = 64, 640, 100
k = int(os.environ.get("DUMMY", "0"))
prio = int(os.environ.get("DUMMY_PRIO", "0"))
dummies = []
for _ in range(k):
    st = torch.cuda.Stream(priority=prio)
    with torch.cuda.stream(st):
        torch.zeros(1, device="cuda").add_(1)
    dummies.append(st)
torch.cuda.synchronize()
rng = np.random.default_rng(0)
# PLAIN=k: k plain contexts (default-class streams; the first one brings the device's copy stream into being) created and used before the ring;
# PLAIN_CLOSE=1 closes them again first
plain = []
for _ in range(int(os.environ.get("PLAIN", "0"))):
    e = cfa.Engine(160, 160, max_batch=2, dtype="bf16")
    e.forward_enqueue(rng.integers(0, 256, (2, 160, 160, 3), dtype=np.uint8)); e.decode_topk(10)
    plain.append(e)
if os.environ.get("PLAIN_CLOSE") == "1":
    for e in plain:
        e.close()
ring = cfa.EngineRing(S, S, depth=2, max_batch=B, dtype=os.environ.get("DTYPE", "bf16"))
e0 = ring.engines[0]
xs = []
for j in range(4):
    x = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    p = e0.device_alloc(x.nbytes); e0.memcpy_h2d(p, x); xs.append(p)
outs = [(e.device_alloc(B * K * 24), e.device_alloc(B * K * 40), e.device_alloc(B * K * 8)) for e in ring.engines]
def step(i):
    e = ring.engines[i % 2]; o = outs[i % 2]
    e.forward_enqueue(xs[i % 4], on_device=True, B=B, in_format=0); e.decode_topk_device(K, o[0], o[1], o[2])
for i in range(10): step(i)
ring.synchronize()
rates = []
for _ in range(9):
    t0 = time.perf_counter()
    for i in range(20): step(i)
    ring.synchronize()
    rates.append(B * 20 / (time.perf_counter() - t0))
names = [(i, w) for i in range(2) for w in (0, 1)]
lab = lambda i, w: ("main%d" % i, "dec%d" % i)[w]
fat = {lab(i, w): [lab(j, v) for (j, v) in names if (j, v) != (i, w) and ring.engines[j].queue_shared(v, ring.engines[i], w + 16)] for (i, w) in names}
print(json.dumps({"plain_contexts": len(plain), "closed": os.environ.get("PLAIN_CLOSE") == "1", "placement": ring.placement, "dummy_streams": k, "blocked_by_fat_kernel_on": fat, "prio": prio, "spread_called": ring.queue_rerolls, "images_per_s": round(float(np.median(rates)), 1)}))
