"""the depthwise tensor of layer4.0-6.0 sub-batched (CF_SUBBATCH=n, experiments build) so that its HBM round trip
can be served by the Infinity Cache.  Prints a digest of the decoded output (must not depend on n) and the forward time.
    CF_LIB=.../libcenterface_hip_exp.so CF_SUBBATCH=16 python tools/subbatch_probe.py [dtype]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import centerface_amd as cfa
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B, S, K = 64, 640, 100
imgs = np.random.default_rng(0).integers(0, 256, (B, S, S, 3), dtype=np.uint8)
d_in = torch.from_numpy(imgs).cuda()
res = {}
for depth in (1, 2):
    engs = [cfa.Engine(S, S, max_batch=B, dtype=dtype) for _ in range(depth)]
    outs = [(torch.empty((B, K, 6), dtype=torch.float32, device="cuda"), torch.empty((B, K, 10), dtype=torch.float32, device="cuda"),
             torch.empty((B, K), dtype=torch.int64, device="cuda")) for _ in engs]
    def step(i):
        e, o = engs[i % depth], outs[i % depth]
        e.forward_enqueue(d_in.data_ptr(), on_device=True, B=B, in_format=cfa._lib.CF_IN_U8_HWC_BGR)
        e.decode_topk_device(K, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr())
    for i in range(6):
        step(i)
    for e in engs: e.synchronize()
    ts = []
    for rep in range(9):
        t0 = time.perf_counter()
        for i in range(20): step(i)
        for e in engs: e.synchronize()
        ts.append((time.perf_counter() - t0) / 20 * 1e3)
    dig = hashlib.sha256(outs[0][0].cpu().numpy().tobytes() + outs[0][2].cpu().numpy().tobytes()).hexdigest()[:16]
    print("CF_SUBBATCH=%s dtype=%s contexts=%d: %.4f ms per step (median of 9 x 20), %.0f img/s, digest %s" %
          (os.environ.get("CF_SUBBATCH", "0"), dtype, depth, float(np.median(ts)), B / float(np.median(ts)) * 1e3, dig))
    for e in engs: e.close()
