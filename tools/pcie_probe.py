#!/usr/bin/env python3
"""PCIe-inclusive rate of the detect path: the batch starts in HOST memory every step (pageable numpy vs
pinned torch buffer), cf_forward copies it H2D on the context stream, then forward + top-K decode."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa

B, S, K = 64, 640, 100
eng = cfa.Engine(S, S, max_batch=B, dtype="bf16")
rng = np.random.default_rng(0)
x = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
xp = torch.from_numpy(x).pin_memory()
outs = (eng.device_alloc(B * K * 6 * 4), eng.device_alloc(B * K * 10 * 4), eng.device_alloc(B * K * 8))
for name, ptr in (("pageable numpy", x.ctypes.data), ("pinned (torch.pin_memory)", xp.data_ptr())):
    def step():
        eng._chk(eng._L.cf_forward(eng._h, ptr, 0, 0, B)); eng.last_B = B
        eng.decode_topk_device(K, *outs)
    for _ in range(3): step()
    eng.synchronize()
    t0 = time.perf_counter(); n = 20
    for _ in range(n): step()
    eng.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-28s %.3f ms/step -> %.0f img/s  (H2D %.1f MB/step)" % (name, dt * 1e3, B / dt, x.nbytes / 1e6))
