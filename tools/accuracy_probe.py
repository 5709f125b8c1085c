#!/usr/bin/env python3
"""How far is the bf16 throughput mode from the fp32 parity mode on the same inputs?  (Both on the GPU; the fp32
mode itself is tested against the reference's goldens / the oracle.)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa

S, B, K = 640, 16, 100
rng = np.random.default_rng(1)
x = rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)
e32 = cfa.Engine(S, S, max_batch=B, dtype="fp32"); e16 = cfa.Engine(S, S, max_batch=B, dtype="bf16")
e32.forward_enqueue(x); e16.forward_enqueue(x)
h32, h16 = e32.heads(sigmoid_hm=True), e16.heads(sigmoid_hm=True)
for k in ("hm", "hm_sigmoid", "wh", "reg", "lm"):
    d = np.abs(h16[k] - h32[k])
    print("%-10s mean|d| %.4g  p99 %.4g  max %.4g   (mean|ref| %.4g)" % (k, d.mean(), np.percentile(d, 99), d.max(), np.abs(h32[k]).mean()))
d32, _, i32 = e32.decode_topk(K); d16, _, i16 = e16.decode_topk(K)
ov, berr, serr = [], [], []
for b in range(B):
    common = set(i32[b].tolist()) & set(i16[b].tolist())
    ov.append(len(common))
    p32 = {int(i): n for n, i in enumerate(i32[b])}; p16 = {int(i): n for n, i in enumerate(i16[b])}
    for i in common:
        berr.append(np.abs(d32[b, p32[i], :4] - d16[b, p16[i], :4]).max()); serr.append(abs(d32[b, p32[i], 4] - d16[b, p16[i], 4]))
print("top-%d index overlap per image: mean %.1f min %d;  matched boxes: mean |d coord| %.4f map px (max %.3f), mean |d score| %.5f"
      % (K, np.mean(ov), min(ov), np.mean(berr), np.max(berr), np.mean(serr)))
