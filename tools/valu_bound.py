#!/usr/bin/env python3
"""'VALU-issue-bound at X %' made checkable: per kernel, DYNAMIC instruction counts from rocprofv3 PMC
(SQ_INSTS_VALU, SQ_INSTS_VALU_TRANS_F32, SQ_INSTS_MFMA, SQ_INSTS_LDS, SQ_INSTS_VMEM; wave-instructions per launch)
x measured issue costs (profiles/r01_valu_microbench.md: cycles per wave64 instruction per SIMD) against the
measured kernel duration (rocprofv3 --kernel-trace --stats of the same session).

The mix of the non-transcendental VALU instructions (plain VOP1/VOP2, VOP3, packed, dot2c, cvt_pk) is taken from the
kernel's ISA (hipcc --save-temps; static histogram of the function body -- these kernels are unrolled straight-line
code inside one chunk loop, so the static mix is the dynamic one to a few percent).

    python tools/valu_bound.py --inst gpurun_out/r02b/prof_inst/i_counter_collection.csv \
        --stats gpurun_out/r02b/prof_k/k_kernel_stats.csv --asm /tmp/asm --out profiles/r02b_valu_bound.md
"""
import argparse
import collections
import csv
import glob
import os
import re
import subprocess

COST = {"trans": 9.45, "pk": 5.65, "dot": 4.53, "vop3": 4.5, "plain": 3.1}     # cycles / wave-instruction / SIMD
LDS_ISSUE, VMEM_ISSUE, MFMA_ISSUE = 4.0, 4.0, 4.0                             # issue slots (the pipes themselves run beside the VALU)
# VALU-issue time an MFMA takes away when it is interleaved with transcendental work on the same SIMD
# (tools/ub_mfma_coissue_probe.hip, profiles/r03_mfma_coissue.md): the matrix pipe overlaps only partly with VALU issue
MFMA_COST = {"4x4x4": 4.5, "16x16": 12.0, "32x32": 22.0}
SIMDS, CLK = 1024, 2.4e9

TRANS = ("v_exp_", "v_rcp_", "v_log_", "v_sqrt_", "v_rsq_", "v_sin_", "v_cos_")


def classify(m):
    if m.startswith(TRANS):
        return "trans"
    if m.startswith("v_pk_"):
        return "pk"
    if m.startswith("v_dot"):
        return "dot"
    if m.startswith(("v_cvt_pk", "v_permlane", "v_mad_", "v_fma_", "v_lshl_add", "v_add3", "v_cndmask_b32_e64", "v_mul_lo", "v_mul_hi")) or m.endswith("_e64"):
        return "vop3"
    return "plain"


def demangle(sym):  # noqa
    sym = sym.strip()
    try:
        return subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
    except Exception:
        return sym


def static_mix(asm_dir):
    mixes = {}
    for path in glob.glob(os.path.join(asm_dir, "*gfx950.s")):
        cur, hist = None, None
        for line in open(path):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                cur, hist = m.group(1), collections.Counter()
                continue
            if cur and "s_endpgm" in line:
                mixes[demangle(cur)] = hist
                cur = None
                continue
            if cur:
                t = line.strip().split()
                if t and t[0].startswith("v_mfma"):
                    hist["mfma:" + ("4x4x4" if "_4x4x4" in t[0] else "16x16" if "_16x16" in t[0] else "32x32")] += 1
                elif t and t[0].startswith("v_"):
                    hist[classify(t[0])] += 1
    return mixes


def counters(path):
    per, name = collections.defaultdict(lambda: collections.defaultdict(float)), {}
    for r in csv.DictReader(open(path)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        name[r["Dispatch_Id"]] = r["Kernel_Name"]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d, v in per.items():
        for c, x in v.items():
            agg[name[d]][c].append(x)
    return {k: {c: sum(x) / len(x) for c, x in v.items()} for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inst", required=True)
    ap.add_argument("--stats", required=True)
    ap.add_argument("--asm", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--json", help="also write the per-kernel counts / costs / bounds as JSON (bench.py reads the newest profiles/*_valu_counts.json)")
    a = ap.parse_args()
    mixes = static_mix(a.asm)
    cnt = counters(a.inst)
    dur = {r["Name"]: float(r["AverageNs"]) for r in csv.DictReader(open(a.stats))}
    lines = ["# VALU-issue bound per kernel (B = 64, 640x640, bf16)", "",
             "Dynamic wave-instruction counts per launch (rocprofv3 --pmc SQ_INSTS_*) x measured issue cost per wave64",
             "instruction per SIMD (transcendental %.2f, packed %.2f, dot2c %.2f, VOP3 / cvt_pk %.1f, plain VOP1/2 %.1f cycles;" %
             (COST["trans"], COST["pk"], COST["dot"], COST["vop3"], COST["plain"]),
             "LDS / VMEM instructions one %.0f-cycle issue slot each; an MFMA 4.5 (4x4x4) / 12 (16x16) / 22 (32x32) cycles of VALU issue) over %d SIMDs at %.1f GHz, against the" % (LDS_ISSUE, SIMDS, CLK / 1e9),
             "measured duration.  `bound/measured` near 1 = the kernel runs at its own instruction-issue bound (rocm-smi shows",
             "sclk 2375-2382 MHz and ~1225 W of the 1400 W cap while bench.py runs: the clock is not the gap).", "",
             "| kernel | M VALU | of which trans | M MFMA | M LDS | avg cost non-trans | issue-bound us | measured us | bound / measured |",
             "|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
    jout = {}
    for k, c in sorted(cnt.items(), key=lambda kv: -dur.get(kv[0], 0)):
        if "cf::" not in k or k not in dur:
            continue
        mix = mixes.get(k)
        mfc = MFMA_ISSUE
        if mix:
            nt = {cl: n for cl, n in mix.items() if cl != "trans" and not cl.startswith("mfma:")}
            avg = sum(COST[cl] * n for cl, n in nt.items()) / max(1, sum(nt.values()))
            mm = {cl[5:]: n for cl, n in mix.items() if cl.startswith("mfma:")}
            if mm:
                mfc = sum(MFMA_COST[t] * n for t, n in mm.items()) / sum(mm.values())
        else:
            avg = 4.0
        valu, tr = c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_VALU_TRANS_F32", 0)
        mf, lds, vm = c.get("SQ_INSTS_MFMA", 0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_VMEM", 0)
        cyc = tr * COST["trans"] + (valu - tr - mf) * avg + lds * LDS_ISSUE + vm * VMEM_ISSUE + mf * mfc
        bound_us = cyc / SIMDS / CLK * 1e6
        jout[k] = {"valu": valu, "trans": tr, "mfma": mf, "lds": lds, "vmem": vm, "avg_cost_nontrans": round(avg, 3), "mfma_cost": round(mfc, 2),
                   "issue_cycles": cyc, "bound_us": round(bound_us, 2), "rocprof_avg_us": round(dur[k] / 1e3, 2)}
        lines.append("| `%s` | %.2f | %.2f | %.2f | %.2f | %.2f | %.1f | %.1f | %.2f |" % (
            k, valu / 1e6, tr / 1e6, mf / 1e6, lds / 1e6, avg, bound_us, dur[k] / 1e3, bound_us / (dur[k] / 1e3)))
    open(a.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    if a.json:
        import json
        json.dump({"workload": "B=64 640x640 bf16 forward + top-100 decode, one launch of each kernel (tools/profile_ops.py)",
                   "cost_cycles_per_wave_instruction": dict(COST, lds=LDS_ISSUE, vmem=VMEM_ISSUE, mfma=MFMA_COST),
                   "simds": SIMDS, "clock_hz": CLK, "kernels": jout}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
