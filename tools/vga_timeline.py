import os, sys, time, json
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import centerface_amd as cfa
from centerface_amd import centerface as c
T = {}
def wrapf(mod, name, key):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k)
        if callable(r):
            def w():
                t2 = time.perf_counter(); r(); T[key + "_wait"] = T.get(key + "_wait", 0.0) + time.perf_counter() - t2
            T[key] = T.get(key, 0.0) + time.perf_counter() - t
            return w
        T[key] = T.get(key, 0.0) + time.perf_counter() - t; return r
    setattr(mod, name, g)
wrapf(c, "_stage_copy_begin", "stage_begin")
def wrapm(obj, name, key):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); T[key] = T.get(key, 0.0) + time.perf_counter() - t; return r
    setattr(obj, name, g)
wrapm(c.Engine, "forward_enqueue", "enqueue")
wrapm(c.Engine, "forward_resized_enqueue", "enqueue_resized")
wrapm(c.CenterFaceBuckets, "_staging", "staging_view")
rng = np.random.default_rng(0)
shapes = [(480, 640), (640, 480), (640, 640), (448, 640), (640, 416)]
imgs = [rng.integers(0, 256, shapes[i % 5] + (3,), dtype=np.uint8) for i in range(128)]
pool = cfa.CenterFaceBuckets(dtype="bf16", max_batch=32, max_buckets=8)
pool.detect(imgs); pool.detect(imgs)
T.clear(); N = 5
t0 = time.perf_counter()
for _ in range(N): pool.detect(imgs)
tot = (time.perf_counter() - t0) / N
print(json.dumps({"total_ms": round(tot * 1e3, 3), **{k: round(v / N * 1e3, 3) for k, v in T.items()}}))
# raw host copy rate of the staging step alone
st = np.empty((26, 480, 640, 3), np.uint8)
t0 = time.perf_counter()
for _ in range(10):
    for k in range(26): np.copyto(st[k], imgs[(5 * k) % 128] if imgs[(5*k)%128].shape == (480,640,3) else imgs[0])
print("single-thread copy of one chunk: %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
