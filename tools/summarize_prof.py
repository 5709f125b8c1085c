#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/) into the tracked summaries under profiles/.

    python tools/summarize_prof.py --tag r01 --stats gpurun_out/prof_k/k_kernel_stats.csv \
        --fetch gpurun_out/prof_fetch/f_counter_collection.csv \
        --write gpurun_out/prof_write/w_counter_collection.csv [--ops gpurun_out/ops.json]

Writes profiles/<tag>_kernel_stats.csv (verbatim copy of rocprofv3 --kernel-trace --stats),
profiles/<tag>_summary.md and profiles/traffic_<tag>.json (per-kernel HBM bytes per launch).

HBM traffic follows /opt/skills/guides/MI355X_MICROARCH.md section HBM: FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE --pmc passes (TCC slot limits), are in KiB, and on gfx950 FETCH_SIZE counts
128-byte requests as 64 bytes for wide coalesced streaming reads, so the read side is DOUBLED;
WRITE_SIZE is taken as reported (uncalibrated per the guide).  Infinity-Cache hits are included in
these fabric-side counters, so `traffic` is an upper bound on true HBM bytes.
"""
import argparse
import csv
import json
import os
import shutil
from collections import defaultdict

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel_counter(path, counter):
    per_dispatch = defaultdict(float)
    name_of = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            d = row["Dispatch_Id"]
            per_dispatch[d] += float(row["Counter_Value"])
            name_of[d] = row["Kernel_Name"]
    agg = defaultdict(lambda: [0.0, 0])
    for d, v in per_dispatch.items():
        a = agg[name_of[d]]
        a[0] += v
        a[1] += 1
    return {k: (v[0] / v[1], v[1]) for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--stats", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--ops", help="tools/profile_ops.py --json output (per-launch table)")
    ap.add_argument("--bench", help="bench.py JSON line measured in the same run")
    ap.add_argument("--stats2", help="kernel stats of the default two-context run")
    ap.add_argument("--lds", help="counter_collection.csv of the SQ_LDS_* pass")
    a = ap.parse_args()
    out = os.path.join(REPO, "profiles")
    os.makedirs(out, exist_ok=True)
    shutil.copyfile(a.stats, os.path.join(out, "%s_kernel_stats.csv" % a.tag))
    rows = list(csv.DictReader(open(a.stats)))
    fetch = per_kernel_counter(a.fetch, "FETCH_SIZE") if a.fetch else {}
    write = per_kernel_counter(a.write, "WRITE_SIZE") if a.write else {}
    traffic = {}
    for k in set(fetch) | set(write):
        f_kib = fetch.get(k, (0.0, 0))[0]
        w_kib = write.get(k, (0.0, 0))[0]
        traffic[k] = {"fetch_kib_raw_per_launch": f_kib, "write_kib_raw_per_launch": w_kib,
                      "hbm_bytes_per_launch": (2.0 * f_kib + w_kib) * 1024.0,
                      "launches_sampled": max(fetch.get(k, (0, 0))[1], write.get(k, (0, 0))[1])}
    with open(os.path.join(out, "traffic_%s.json" % a.tag), "w") as f:
        json.dump(traffic, f, indent=1, sort_keys=True)
    lines = ["# rocprofv3 summary `%s`" % a.tag, "",
             "Note: the decode kernels (`peak_collect_kernel`, `topk_select_kernel`) run on the context's second stream "
             "underneath the next forward (device-output decode in bench.py), so their durations here are stretched by "
             "sharing the chip; timed alone they are in the per-launch table below.", "",
             "Source: `rocprofv3 --kernel-trace --stats -- python bench.py --depth 1 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-extras` "
             "(ONE context, as bench.py's roofline block times the kernels; with the default ring of two contexts kernels of two "
             "batches share the chip and every duration stretches -- second table) "
             "(kernel durations) and two `--pmc` passes over `tools/profile_ops.py` (FETCH_SIZE, WRITE_SIZE; "
             "read side doubled per the gfx950 note in MI355X_MICROARCH.md).", "",
             "| kernel | calls | avg us | % GPU time | HBM MB/launch (PMC) |", "|---|---:|---:|---:|---:|"]
    for r in rows:
        name = r["Name"]
        if not name.startswith(("void cf::", "cf::")):
            continue
        t = traffic.get(name)
        lines.append("| `%s` | %s | %.1f | %s | %s |" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"],
                                                       ("%.1f" % (t["hbm_bytes_per_launch"] / 1e6)) if t else "-"))
    if a.stats2:
        shutil.copyfile(a.stats2, os.path.join(out, "%s_kernel_stats_two_contexts.csv" % a.tag))
        lines += ["", "Two contexts (bench.py default, `--depth 2`): kernels of two batches overlap, durations are per kernel while sharing the chip:", "",
                  "| kernel | calls | avg us | % GPU time |", "|---|---:|---:|---:|"]
        for r in csv.DictReader(open(a.stats2)):
            if r["Name"].startswith(("void cf::", "cf::")):
                lines.append("| `%s` | %s | %.1f | %s |" % (r["Name"], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
    if a.lds:
        agg = {}
        for r in csv.DictReader(open(a.lds)):
            agg.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], 0.0)
            agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
        lines += ["", "LDS bank conflicts (`--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS` over `tools/profile_ops.py`):", "",
                  "| kernel | conflict cycles / LDS active cycles | LDS cycles per LDS instruction |", "|---|---:|---:|"]
        for k, v in agg.items():
            if k.startswith(("void cf::", "cf::")) and v.get("SQ_INSTS_LDS", 0) > 0:
                lines.append("| `%s` | %.2f | %.1f |" % (k, v.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, v.get("SQ_LDS_IDX_ACTIVE", 0)),
                                                      v.get("SQ_LDS_IDX_ACTIVE", 0) / v["SQ_INSTS_LDS"]))
    if a.bench:
        lines += ["", "bench.py line of the same session:", "", "```", open(a.bench).read().strip(), "```"]
    if a.ops:
        ops = json.load(open(a.ops))
        lines += ["", "Per-launch table (HIP events on the ctx stream, B=%d, %s):" % (ops["batch"], ops["dtype"]), "",
                  "| layer | kind | ms | algorithmic GB/s | TFLOP/s |", "|---|---|---:|---:|---:|"]
        for o in ops["ops"]:
            lines.append("| %s | %s | %.4f | %.0f | %.1f |" % (o["name"], o["kind"], o["ms"], o["GBps"], o["TFLOPs"]))
        lines.append("")
        lines.append("forward back-to-back: %.3f ms" % ops["forward_ms"])
    with open(os.path.join(out, "%s_summary.md" % a.tag), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main()
