#!/usr/bin/env python3
"""Per-launch profile of one forward (+decode) with HIP events on the ctx stream: name, kernel
symbol, ms, algorithmic GB/s, TFLOP/s.  Run on the GPU box:

    python tools/profile_ops.py [--batch 64] [--size 640] [--dtype bf16] [--reps 5] [--json out.json]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--topk", type=int, default=100)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--collapse", action="store_true")
    ap.add_argument("--no-fuse", action="store_true", help="three-kernel MBConv path (CF_FLAG_NO_FUSE)")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (a.batch, a.size, a.size, 3), dtype=np.uint8)
    eng = cfa.Engine(a.size, a.size, max_batch=a.batch, dtype=a.dtype, collapse_heads=True if a.collapse else None, fuse=not a.no_fuse)
    d_in = eng.device_alloc(img.nbytes)
    eng.memcpy_h2d(d_in, img)
    fmt = cfa._lib.CF_IN_U8_HWC_BGR
    for _ in range(2):
        eng.profile_forward(d_in, on_device=True, B=a.batch, in_format=fmt, K=a.topk)
    acc = None
    for _ in range(a.reps):
        recs = eng.profile_forward(d_in, on_device=True, B=a.batch, in_format=fmt, K=a.topk)
        if acc is None:
            acc = recs
        else:
            for x, y in zip(acc, recs):
                x["ms"] += y["ms"]
    tot = 0.0
    print("%-20s %-6s %9s %9s %8s  %s" % ("layer", "kind", "ms", "GB/s", "TFLOP/s", "kernel"))
    for r in acc:
        r["ms"] /= a.reps
        tot += r["ms"]
        gbs = r["algo_bytes"] / (r["ms"] * 1e-3) / 1e9 if r["ms"] > 0 else 0
        tf = r["flops"] / (r["ms"] * 1e-3) / 1e12 if r["ms"] > 0 else 0
        r["GBps"], r["TFLOPs"] = gbs, tf
        print("%-20s %-6s %9.4f %9.1f %8.2f  %s" % (r["name"], r["kind"], r["ms"], gbs, tf, r["kernel"]))
    print("sum of kernels: %.3f ms  -> %.0f img/s (B=%d)" % (tot, a.batch / (tot * 1e-3), a.batch))
    # un-instrumented back-to-back timing
    eng.forward_enqueue(d_in, on_device=True, B=a.batch, in_format=fmt)
    eng.synchronize()
    eng.event_record(0)
    n = 10
    for _ in range(n):
        eng.forward_enqueue(d_in, on_device=True, B=a.batch, in_format=fmt)
    eng.event_record(1)
    ms = eng.event_elapsed_ms(0, 1) / n
    print("forward back-to-back: %.3f ms -> %.0f img/s" % (ms, a.batch / (ms * 1e-3)))
    if a.json:
        with open(a.json, "w") as f:
            json.dump({"batch": a.batch, "size": a.size, "dtype": a.dtype, "ops": acc, "forward_ms": ms}, f, indent=1)
    eng.device_free(d_in)
    eng.close()


if __name__ == "__main__":
    main()
