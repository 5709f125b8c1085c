#!/usr/bin/env python3
"""Host-side timestamps (ms from the start of the call) of every Engine call inside one CenterFaceBuckets.detect on the configs[3] mix."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa
from centerface_amd import centerface as c
LOG = []
T0 = [0.0]
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); LOG.append((t - T0[0], time.perf_counter() - T0[0], threading.current_thread().name[:10], name)); return r
    setattr(obj, name, g)
ENG = []
def wrap_ev(name, before, after):
    f = getattr(c.Engine, name)
    def g(self, *a, **k):
        if before is not None:
            self.event_record(before)
            self._t_enq = time.perf_counter() - T0[0]
            if self not in ENG: ENG.append(self)
        r = f(self, *a, **k)
        self.event_record(after)
        return r
    setattr(c.Engine, name, g)
if os.environ.get("VGA_EVENTS", "1") == "1":
    for n in ("forward_enqueue", "forward_resized_enqueue", "forward_images_enqueue"):
        wrap_ev(n, 0, 1)
    wrap_ev("decode_threshold_enqueue", None, 2)
    wrap_ev("_upload_addrs", 0, 3)
    wrap_ev("forward_uploaded", None, 1)
for n in ("forward_enqueue", "forward_resized_enqueue", "forward_images_enqueue", "_upload_addrs", "forward_uploaded", "decode_threshold_enqueue", "decode_threshold", "set_rescale"):
    wrap(c.Engine, n)
rng = np.random.default_rng(0)
shapes = [(480, 640), (640, 480), (640, 640), (448, 640), (640, 416)]
imgs = [rng.integers(0, 256, shapes[i % 5] + (3,), dtype=np.uint8) for i in range(128)]
if os.environ.get("VGA_PINNED", "1") == "1":
    p = []
    for im in imgs:
        a = cfa.pinned_empty(im.shape); a[...] = im; p.append(a)
    imgs = p
pool = cfa.CenterFaceBuckets(dtype="bf16", max_batch=32, max_buckets=8)
for _ in range(4):
    pool.detect(imgs)
LOG.clear(); ENG.clear(); T0[0] = time.perf_counter()
pool.detect(imgs)
end = time.perf_counter() - T0[0]
for e in ENG:
    print("engine %dx%d  enqueued at %.3f ms (host)   device: forward done +%.3f ms, decode done +%.3f ms  -> decode done at ~%.3f" % (
        e.H, e.W, e._t_enq * 1e3, e.event_elapsed_ms(0, 1), e.event_elapsed_ms(0, 2), e._t_enq * 1e3 + e.event_elapsed_ms(0, 2)))
for a, b, th, n in sorted(LOG):
    if n != "set_rescale" or b - a > 0.00005:
        print("%.3f - %.3f  %-10s %s" % (a * 1e3, b * 1e3, th, n))
print("detect end %.3f" % (end * 1e3))
