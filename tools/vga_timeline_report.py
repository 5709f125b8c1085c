#!/usr/bin/env python3
"""Per-stream timeline of the last CenterFaceBuckets.detect call in a rocprofv3 --kernel-trace --memory-copy-trace database
(rocpd .db) of tools/vga_timeline.py: when the copies of each chunk run, when its kernels run, what the device waits for."""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
K = list(cur.execute("select name, start, end, stream_id from kernels order by start"))
C = list(cur.execute("select start, end, size, stream_id from rocpd_memory_copy order by start"))
ev = sorted([(s, e, "K", st, n.split("(")[0].replace("void cf::", "").replace("cf::", "")[:44]) for n, s, e, st in K] +
            [(s, e, "C", st, sz) for s, e, sz, st in C])
calls, cur_, last = [], [], None
for x in ev:
    if last is not None and x[0] - last > 20e6:
        calls.append(cur_); cur_ = []
    cur_.append(x); last = max(last or 0, x[1])
calls.append(cur_)
c = calls[-1]
t0 = c[0][0]
print("calls", [len(k) for k in calls], "| last call: span %.3f ms" % ((max(x[1] for x in c) - t0) / 1e6))
bys = defaultdict(list)
for x in c:
    bys[(x[2], x[3])].append(x)
for k, l in sorted(bys.items(), key=lambda kv: kv[1][0][0]):
    mb = sum(x[4] for x in l if x[2] == "C") / 1e6
    print("%s stream %-3s n %3d  first %.3f  last end %.3f  busy %.3f ms%s" % (k[0], k[1], len(l), (l[0][0] - t0) / 1e6, (max(x[1] for x in l) - t0) / 1e6,
                                                                       sum(x[1] - x[0] for x in l) / 1e6, "  %.1f MB" % mb if mb else ""))
if len(sys.argv) > 2:
    for x in c:
        if x[2] == "K" or x[4] < 100000:
            print("%.3f %.3f %s %s %s" % ((x[0] - t0) / 1e6, (x[1] - t0) / 1e6, x[2], x[3], x[4]))
