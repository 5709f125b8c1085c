#!/usr/bin/env python3
"""Timing probe: one context at batch B against N contexts at batch B/N whose forwards are enqueued back to back on
their own streams (kernels of different contexts may overlap on the GPU: HBM-bound GEMMs / up3+heads under the
VALU-bound fused blocks of another sub-batch, and the tail of every kernel under the head of another).

    python tools/dual_stream_probe.py [--batch 64] [--size 640] [--splits 2 4] [--steps 20] [--repeats 15]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--topk", type=int, default=100)
    ap.add_argument("--splits", type=int, nargs="*", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=15)
    ap.add_argument("--alternate", type=int, nargs="*", default=[], help="N full-batch contexts, step k runs on context k %% N")
    a = ap.parse_args()
    B, S, K = a.batch, a.size, a.topk
    fmt = cfa._lib.CF_IN_U8_HWC_BGR
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    for n in a.splits:
        b = B // n
        engs = [cfa.Engine(S, S, max_batch=b, dtype="bf16") for _ in range(n)]
        d_in = engs[0].device_alloc(img.nbytes)
        engs[0].memcpy_h2d(d_in, img)
        outs = []
        for e in engs:
            outs.append((e.device_alloc(b * K * 6 * 4), e.device_alloc(b * K * 10 * 4), e.device_alloc(b * K * 8)))
        per = img.nbytes // n

        def step():
            for i, e in enumerate(engs):
                e.forward_enqueue(d_in + i * per, on_device=True, B=b, in_format=fmt)
                e.decode_topk_device(K, *outs[i])

        def fence():
            for e in engs:
                e.synchronize()
        for _ in range(5):
            step()
        fence()
        ts = []
        for _ in range(a.repeats):
            fence()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                step()
            fence()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        med = ts[len(ts) // 2]
        print("splits=%d (batch %d each): median %.3f ms/step  %.0f img/s  (min %.3f max %.3f)" %
              (n, b, med / a.steps * 1e3, B * a.steps / med, ts[0] / a.steps * 1e3, ts[-1] / a.steps * 1e3), flush=True)
        for e in engs:
            e.close()
    for n in a.alternate:
        alternate(a, n)


def alternate(a, n):
    B, S, K = a.batch, a.size, a.topk
    fmt = cfa._lib.CF_IN_U8_HWC_BGR
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
    engs = [cfa.Engine(S, S, max_batch=B, dtype="bf16") for _ in range(n)]
    d_in = engs[0].device_alloc(img.nbytes)
    engs[0].memcpy_h2d(d_in, img)
    outs = [(e.device_alloc(B * K * 6 * 4), e.device_alloc(B * K * 10 * 4), e.device_alloc(B * K * 8)) for e in engs]
    k = [0]

    def step():
        i = k[0] % n
        k[0] += 1
        engs[i].forward_enqueue(d_in, on_device=True, B=B, in_format=fmt)
        engs[i].decode_topk_device(K, *outs[i])

    def fence():
        for e in engs:
            e.synchronize()
    for _ in range(6):
        step()
    ts = []
    for _ in range(a.repeats):
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        fence()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    print("alternate=%d (batch %d each): median %.3f ms/step  %.0f img/s  (min %.3f max %.3f)" %
          (n, B, med / a.steps * 1e3, B * a.steps / med, ts[0] / a.steps * 1e3, ts[-1] / a.steps * 1e3), flush=True)
    for e in engs:
        e.close()


if __name__ == "__main__":
    main()
