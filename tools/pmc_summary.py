#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: counters as % of SQ_WAVE_CYCLES."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
disp = defaultdict(dict); name = {}
for r in rows:
    d = r['Dispatch_Id']; disp[d][r['Counter_Name']] = disp[d].get(r['Counter_Name'], 0) + float(r['Counter_Value']); name[d] = r['Kernel_Name']
agg = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for d, c in disp.items():
    for k, v in c.items(): agg[name[d]][k] += v
    cnt[name[d]] += 1
ctrs = sorted({k for a in agg.values() for k in a})
print("%-72s %5s " % ("kernel", "n") + " ".join("%14s" % c[-14:] for c in ctrs))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', kv[1].get(ctrs[0], 0))):
    if 'cf::' not in k: continue
    wc = a.get('SQ_WAVE_CYCLES')
    vals = []
    for c in ctrs:
        v = a[c] / cnt[k]
        vals.append("%14.3g" % v if (wc is None or c == 'SQ_WAVE_CYCLES') else "%13.1f%%" % (100 * a[c] / wc))
    print("%-72s %5d " % (k.replace('void cf::', '').replace('unsigned short', 'bf16')[:72], cnt[k]) + " ".join(vals))
