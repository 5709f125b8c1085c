#!/usr/bin/env python3
"""Soak run on the GPU box: thousands of mixed calls (device / host / resized inputs, both decoders on both streams,
the RCCL gather at world 1, weight reloads, two batch sizes) on one context, every result compared with the first
occurrence of the same call on the same input.  `python tools/soak.py [--seconds 60]`"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=60.0); a = ap.parse_args()
    import torch
    rng = np.random.default_rng(0)
    H, W, B = 160, 224, 6
    eng = cfa.Engine(H, W, max_batch=B, dtype="bf16")
    comm = cfa.distributed.Comm(eng, 0, 1, cfa.distributed.unique_id())
    full = [rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8) for _ in range(3)]
    small = [rng.integers(0, 256, (B - 2, 97, 131, 3), dtype=np.uint8) for _ in range(2)]
    dev = [torch.from_numpy(x).cuda() for x in full]
    d_dev = torch.empty((B, 40, 6), dtype=torch.float32, device="cuda")
    seen, n, t0 = {}, 0, time.time()
    sd = cfa.weights.synthetic_state_dict(0)
    while time.time() - t0 < a.seconds:
        op = int(rng.integers(0, 7)); k = int(rng.integers(0, 3))
        if op == 0:
            eng.forward_enqueue(full[k]); key = ("h", k)
        elif op == 1:
            eng.forward_enqueue(dev[k].data_ptr(), on_device=True, B=B, in_format=0); key = ("h", k)
        elif op == 2:
            eng.forward_resized_enqueue(small[k % 2]); key = ("r", k % 2)
        elif op == 3:
            eng.forward_enqueue(full[k][:2]); key = ("p", k)
        elif op == 4:
            eng.forward_enqueue(full[k]); eng.decode_topk_device(40, d_dev.data_ptr()); eng.forward_enqueue(full[(k + 1) % 3])
            eng.synchronize(); got = d_dev.cpu().numpy(); key2 = ("dev", k)
            if key2 in seen: assert np.array_equal(seen[key2], got), key2
            seen.setdefault(key2, got); key = ("h", (k + 1) % 3)
        elif op == 5:
            eng.forward_enqueue(full[k]); rec = comm.gather_topk(40); key2 = ("rec", k)
            if key2 in seen: assert np.array_equal(seen[key2], rec), key2
            seen.setdefault(key2, rec); key = ("h", k)
        else:
            eng.load_state_dict(sd); eng.forward_enqueue(full[k]); key = ("h", k)
        d, l, i = eng.decode_topk(40)
        t = eng.decode_threshold(0.3, 0.3, 128)
        cur = (d, l, i, [x for x, _ in t])
        if key in seen:
            ref = seen[key]
            assert np.array_equal(cur[0], ref[0]) and np.array_equal(cur[1], ref[1]) and np.array_equal(cur[2], ref[2]), (key, n)
            assert all(np.array_equal(x, y) for x, y in zip(cur[3], ref[3])), (key, n)
        seen.setdefault(key, cur)
        n += 1
    comm.close(); eng.close()
    print("soak ok: %d mixed calls in %.0f s, %d distinct results, all reproducible" % (n, time.time() - t0, len(seen)))


if __name__ == "__main__":
    main()
