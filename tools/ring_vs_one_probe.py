"""Forward-only and forward + decode step times on the ring of two contexts and on one context (B = 64, 640x640, bf16): what the second batch in flight buys.\n    gpurun -- python3 tools/ring_vs_one_probe.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import centerface_amd as cfa
import bench
B, S, K = 64, 640, 100
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
xs = [torch.from_numpy(rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)).to(dev) for _ in range(4)]
for depth in (2, 1):
    ring = cfa.EngineRing(S, S, depth=depth, max_batch=B, dtype="bf16", device=0)
    outs = [{"dets": torch.empty((B, K, 6), dtype=torch.float32, device=dev), "lms": torch.empty((B, K, 10), dtype=torch.float32, device=dev),
             "inds": torch.empty((B, K), dtype=torch.int64, device=dev)} for _ in ring.engines]
    fmt = cfa._lib.CF_IN_U8_HWC_BGR
    k = [0]
    def step_nodec():
        i = k[0] % depth; src = xs[k[0] % 4].data_ptr(); k[0] += 1
        ring.engines[i].forward_enqueue(src, on_device=True, B=B, in_format=fmt)
    st = bench.make_step(cfa, ring.engines, [t.data_ptr() for t in xs], B, K, outs)
    def fence():
        for e in ring.engines: e.synchronize()
        torch.cuda.synchronize()
    for name, f in (("fwd+decode", st), ("fwd only", step_nodec), ("fwd+decode", st), ("fwd only", step_nodec)):
        for _ in range(6): f()
        w = bench.time_windows(f, fence, 20, 9)
        print("depth %d %-11s %.4f ms/step  %.0f img/s" % (depth, name, float(np.median(w)) / 20 * 1e3, B * 20 / float(np.median(w))))
    ring.close()
