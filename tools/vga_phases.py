"""Host-side phase breakdown of CenterFaceBuckets.detect on the configs[3] VGA mix (enqueue / decode+sync / postprocess, ms per call)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa
from centerface_amd import centerface as c
T = {}
def wrap(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[key] = T.get(key, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
pass
wrap(c.Engine, "forward_enqueue", "enqueue")
wrap(c.Engine, "forward_resized_enqueue", "enqueue_resized")
wrap(c.Engine, "forward_images_enqueue", "enqueue_images")
wrap(c.Engine, "decode_threshold", "decode_threshold")
wrap(c.CenterFace, "_postprocess_many", "postprocess")
rng = np.random.default_rng(0)
shapes = [(480, 640), (640, 480), (640, 640), (448, 640), (640, 416)]
imgs = [rng.integers(0, 256, shapes[i % 5] + (3,), dtype=np.uint8) for i in range(128)]
pool = cfa.CenterFaceBuckets(dtype="bf16", max_batch=32, max_buckets=8)
if os.environ.get("VGA_PINNED", "1") == "1":
    pimgs = []
    for im in imgs:
        a = cfa.pinned_empty(im.shape); a[...] = im; pimgs.append(a)
    imgs = pimgs
pool.detect(imgs); pool.detect(imgs)
T.clear()
t0 = time.perf_counter()
N = 5
for _ in range(N): pool.detect(imgs)
tot = (time.perf_counter() - t0) / N
print(json.dumps({"total_ms": round(tot * 1e3, 3), **{k: round(v / N * 1e3, 3) for k, v in T.items()}}))
