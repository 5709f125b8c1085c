#!/usr/bin/env python3
"""Which streams of an EngineRing share a hardware queue (cf_streams_share_queue_ex): main / decode of every context + the copy stream."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import centerface_amd as cfa
depth = int(os.environ.get("DEPTH", "2"))
ring = cfa.EngineRing(640, 640, depth=depth, max_batch=64, dtype="bf16")
names = [(i, w) for i in range(depth) for w in (0, 1)] + [(0, 2)]
label = lambda i, w: ("main%d" % i, "dec%d" % i, "copy")[w] if w < 2 else "copy"
rows = {}
for (i, w) in names:
    rows[label(i, w)] = [label(j, v) for (j, v) in names if (j, v) != (i, w) and ring.engines[i].queue_shared(w, ring.engines[j], v)]
pipes = {}
for (i, w) in names[:-1]:
    pipes[label(i, w)] = [label(j, v) for (j, v) in names[:-1] if (j, v) != (i, w) and ring.engines[i].queue_shared(w, ring.engines[j], v + 16)]
print(json.dumps({"depth": depth, "placed_afresh": bool(ring.queue_rerolls), "shares_queue_with": rows, "waits_for_a_non_resident_grid_on": pipes}))
