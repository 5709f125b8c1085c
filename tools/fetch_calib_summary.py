#!/usr/bin/env python3
"""Summarise tools/ub_fetch_calib.hip runs under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE: bytes actually moved per
counter unit, per access width.    python tools/fetch_calib_summary.py <fetch_counter_collection.csv> <write_...csv>"""
import csv
import sys
from collections import defaultdict

NBYTES = 1 << 30
KNOWN = {"read_stem_patch": 78643200}


def load(path, counter):
    per = defaultdict(lambda: defaultdict(float))
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            per[row["Kernel_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in per.items()}


def main():
    fetch = load(sys.argv[1], "FETCH_SIZE")
    write = load(sys.argv[2], "WRITE_SIZE")
    print("| kernel | bytes moved (known) | counter (KiB) | bytes per counted byte |")
    print("|---|---:|---:|---:|")
    for name, kib in sorted(fetch.items()):
        if "read" not in name:
            continue
        known = next((v for k, v in KNOWN.items() if k in name), NBYTES)
        print("| FETCH_SIZE `%s` | %d | %.0f | %.3f |" % (name, known, kib, known / (kib * 1024.0)))
    for name, kib in sorted(write.items()):
        if "write_k" not in name:
            continue
        print("| WRITE_SIZE `%s` | %d | %.0f | %.3f |" % (name, NBYTES, kib, NBYTES / (kib * 1024.0)))


if __name__ == "__main__":
    main()
