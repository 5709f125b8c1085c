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
# The second table: ARCHITECTURAL issue rates in shader cycles (MI355X_MICROARCH.md: v_fma_f32 2 cycles = the 157 TFLOP/s vector
# peak; transcendentals quarter rate = 8; packed fp32 / dot2 move two elements per lane = 4), VALU instructions only, at the full
# 2.4 GHz.  tools/ub_clock_probe.hip (profiles/r05_valu_clock_probe.md) reconciles the two: a dense v_fma_f32 stream issues at 2.28
# shader cycles per instruction (loop and dependency overhead over the architectural 2.0) and pulls the shader clock down to
# 1.94 GHz (power cap), which is 2.8 'cycles' of wall time at a nominal 2.4 GHz -- the COST table above; transcendentals measure
# 8.1 shader cycles at 2.38 GHz.  The architectural table is a floor no sustained instruction stream reaches on this chip; the
# measured table is what a stream of that mix costs in wall time.
COST_HW = {"trans": 8.0, "pk": 4.0, "dot": 4.0, "vop3": 2.0, "plain": 2.0}

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
             "Second table (`hw bound`): VALU instructions only at ARCHITECTURAL rates (transcendental 8, packed / dot2 4, everything else 2 shader cycles,",
             "2.4 GHz): the floor the hardware guide's 157 TFLOP/s vector peak implies.  tools/ub_clock_probe.hip shows why no sustained stream reaches it:",
             "a dense v_fma_f32 stream issues at 2.28 shader cycles and holds only 1.94 GHz under the power cap (= 2.8 nominal cycles of wall time).", "",
             "| kernel | M VALU | of which trans | M MFMA | M LDS | avg cost non-trans | issue-bound us | measured us | bound / measured | hw bound us | hw bound / measured |",
             "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    jout = {}
    for k, c in sorted(cnt.items(), key=lambda kv: -dur.get(kv[0], 0)):
        if "cf::" not in k or k not in dur:
            continue
        mix = mixes.get(k)
        mfc = MFMA_ISSUE
        avg_hw = 2.0
        if mix:
            nt = {cl: n for cl, n in mix.items() if cl != "trans" and not cl.startswith("mfma:")}
            avg = sum(COST[cl] * n for cl, n in nt.items()) / max(1, sum(nt.values()))
            avg_hw = sum(COST_HW[cl] * n for cl, n in nt.items()) / max(1, sum(nt.values()))
            mm = {cl[5:]: n for cl, n in mix.items() if cl.startswith("mfma:")}
            if mm:
                mfc = sum(MFMA_COST[t] * n for t, n in mm.items()) / sum(mm.values())
        else:
            avg = 4.0
        valu, tr = c.get("SQ_INSTS_VALU", 0), c.get("SQ_INSTS_VALU_TRANS_F32", 0)
        mf, lds, vm = c.get("SQ_INSTS_MFMA", 0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_VMEM", 0)
        cyc = tr * COST["trans"] + (valu - tr - mf) * avg + lds * LDS_ISSUE + vm * VMEM_ISSUE + mf * mfc
        bound_us = cyc / SIMDS / CLK * 1e6
        cyc_hw = tr * COST_HW["trans"] + (valu - tr - mf) * avg_hw                    # VALU instructions only, architectural rates
        cyc_hw2 = tr * 8.0 + (valu - tr) * 2.0                                          # VERDICT r04's own arithmetic: 8 / 2 for everything SQ_INSTS_VALU counts
        jout[k] = {"valu": valu, "trans": tr, "mfma": mf, "lds": lds, "vmem": vm, "avg_cost_nontrans": round(avg, 3), "mfma_cost": round(mfc, 2),
                   "issue_cycles": cyc, "bound_us": round(bound_us, 2), "rocprof_avg_us": round(dur[k] / 1e3, 2),
                   "avg_cost_nontrans_hw": round(avg_hw, 3), "bound_hw_us": round(cyc_hw / SIMDS / CLK * 1e6, 2), "bound_hw_8_2_us": round(cyc_hw2 / SIMDS / CLK * 1e6, 2)}
        lines.append("| `%s` | %.2f | %.2f | %.2f | %.2f | %.2f | %.1f | %.1f | %.2f | %.1f | %.2f |" % (
            k, valu / 1e6, tr / 1e6, mf / 1e6, lds / 1e6, avg, bound_us, dur[k] / 1e3, bound_us / (dur[k] / 1e3),
            jout[k]["bound_hw_us"], jout[k]["bound_hw_us"] / (dur[k] / 1e3)))
    open(a.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))
    if a.json:
        import json
        json.dump({"workload": "B=64 640x640 bf16 forward + top-100 decode, one launch of each kernel (tools/profile_ops.py)",
                   "cost_cycles_per_wave_instruction": dict(COST, lds=LDS_ISSUE, vmem=VMEM_ISSUE, mfma=MFMA_COST),
                   "cost_hw_cycles_per_wave_instruction": COST_HW,
                   "simds": SIMDS, "clock_hz": CLK, "kernels": jout}, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
