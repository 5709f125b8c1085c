#!/usr/bin/env python3
"""GPU (bf16 engine) vs bf16-emulating oracle: per-kernel error statistics in units of the test tolerance
(2^-7 |emu| + 2^-8 rms).  Prints a markdown table (committed under profiles/ as the evidence for the bounds
asserted in tests/test_bf16_parity.py)."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import centerface_amd as cfa                      # noqa: E402
from centerface_amd import ops                    # noqa: E402
from oracle import bf16_emulation as E            # noqa: E402
import test_bf16_parity as T                      # noqa: E402


def stats(got, ref):
    ref = ref.numpy() if hasattr(ref, "numpy") else ref
    r = T._ratio(got, ref)
    return "%.2f | %.1e | %.1e | %.1e" % (r.max(), (r > 1).mean(), (r > 0.5).mean(), (r > 0).mean())


def main():
    print("| kernel (shape) | max |d|/tol | frac > tol | frac > tol/2 | frac differing |")
    print("|---|---:|---:|---:|---:|")
    for blk in T.BLOCKS:
        prefix, cin, cout, k, s, h = blk
        rng = np.random.default_rng(sum(ord(ch) for ch in prefix))
        x = T._bf16_normal(rng, (2, cin, h, h), 1.2)
        we, wd, wp = T._w(prefix)
        res = cin == cout and s == 1
        if prefix in E.SPLIT_BLOCKS:
            y = ops.expand_dw(x, we, wd, k, s, dtype="bf16")
            print("| %s expand+dw (2x%dx%dx%d) | %s |" % (prefix, cin, h, h, stats(y, E.expand_dw(torch.from_numpy(x), we, wd, k, s, out_scaled=False))))
            o = ops.conv_pw(y, wp, residual=x if res else None, dtype="bf16")
            print("| %s project | %s |" % (prefix, stats(o, E.pw_op(y, wp, residual=x if res else None))))
        else:
            y = ops.mbconv(x, we, wd, wp, k, s, dtype="bf16")
            print("| %s fused (2x%dx%dx%d) | %s |" % (prefix, cin, h, h, stats(y, E.mbconv_fused(torch.from_numpy(x), we, wd, wp, k, s, res))))
    for size, B in (((640, 640), 2), ((96, 128), 3)):
        H, W = size
        rng = np.random.default_rng(H + 3 * W)
        x = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
        eng = cfa.Engine(H, W, max_batch=B, dtype="bf16")
        g, rec = T._engine_record(eng, x)
        worst = E.check_blockwise(T.SD, g, detail=True)
        for k, v in worst.items():
            print("| engine %dx%d B=%d, teacher-forced %s | %.2f | %.1e | %.1e | %.1e |" % ((H, W, B, k) + tuple(v[:4])))
        eng.close()


if __name__ == "__main__":
    main()
