#!/bin/bash
# gpurun_out/<tag> (tools/capture_profile.sh on the GPU box) -> the tracked summaries under profiles/.  Run in the container:
#   tools/publish_profile.sh r04a
set -e
TAG=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
G=$ROOT/gpurun_out/$TAG
P=$ROOT/profiles
SRC=$ROOT/lightweight-face-detection-centernet_amd/csrc
# ISA of the kernels for the static instruction mix (tools/valu_bound.py)
ASM=/tmp/asm_$TAG; rm -rf $ASM; mkdir -p $ASM
for f in cf_mbconv2 cf_mbconv2_ilp cf_mbconv3 cf_mbconv3_ilp cf_stem0 cf_uphead cf_neck cf_pw cf_decode; do
  EX=""; [[ $f == *_ilp ]] && EX="-mllvm -amdgpu-sched-strategy=max-ilp"; [[ $f == cf_decode ]] && EX="-ffp-contract=off"
  (cd $SRC && hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -I$ROOT/include -I. $EX -S --offload-device-only $f.hip -o $ASM/$f-hip-amdgcn-amd-amdhsa-gfx950.s) &
done; wait
python $ROOT/tools/summarize_prof.py --tag $TAG --stats $G/prof_k/k_kernel_stats.csv --stats2 $G/prof_k2/k_kernel_stats.csv \
    --fetch $G/prof_fetch/f_counter_collection.csv --write $G/prof_write/w_counter_collection.csv --ops $G/ops.json \
    --bench $G/bench.json --lds $G/prof_lds/l_counter_collection.csv > /dev/null
python $ROOT/tools/valu_bound.py --inst $G/prof_inst/i_counter_collection.csv --stats $G/prof_k/k_kernel_stats.csv --asm $ASM \
    --out $P/${TAG}_valu_bound.md --json $P/${TAG}_valu_counts.json > /dev/null
python $ROOT/tools/pmc_summary.py $G/prof_sq/s_counter_collection.csv > $P/${TAG}_pmc_sq.txt
cp $G/prof_k1280/k_kernel_stats.csv $P/${TAG}_1280_b4_kernel_stats.csv
cp $G/bench.json $P/bench_${TAG}.json; cp $G/bench_1280_b4.json $P/bench_${TAG}_1280_b4.json; cp $G/bench_1280_b32.json $P/bench_${TAG}_1280_b32.json
cp $G/vga_buckets.json $P/${TAG}_vga_buckets.json
cp $G/ops_fp32_b64.txt $P/${TAG}_ops_fp32_b64.txt
# tolerance mode
cp $G/ops_split_b64.txt $P/${TAG}_ops_split_b64.txt
cp $G/prof_ksplit/k_kernel_stats.csv $P/${TAG}_split_kernel_stats.csv
python $ROOT/tools/pmc_summary.py $G/prof_sq_split/s_counter_collection.csv > $P/${TAG}_pmc_sq_split.txt
python $ROOT/tools/pmc_summary.py $G/prof_lds_split/l_counter_collection.csv > $P/${TAG}_pmc_lds_split.txt
python - <<PY
import csv, json, collections
def per_kernel(path, counter):
    per, name = collections.defaultdict(float), {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[r["Dispatch_Id"]] += float(r["Counter_Value"]); name[r["Dispatch_Id"]] = r["Kernel_Name"]
    agg = collections.defaultdict(list)
    for d, v in per.items(): agg[name[d]].append(v)
    return {k: sum(v) / len(v) for k, v in agg.items()}
f = per_kernel("$G/prof_fetch_split/f_counter_collection.csv", "FETCH_SIZE"); w = per_kernel("$G/prof_write_split/w_counter_collection.csv", "WRITE_SIZE")
out = {k: {"fetch_kib_raw_per_launch": f.get(k, 0.0), "write_kib_raw_per_launch": w.get(k, 0.0), "hbm_bytes_per_launch": (2.0 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024.0}
       for k in set(f) | set(w) if "cf::" in k}
json.dump(out, open("$P/${TAG}_split_traffic.json", "w"), indent=1, sort_keys=True)   # not traffic_*.json: bench.py reads the newest of those for the bf16 headline
print("split-mode PMC traffic, MB per batch:", round(sum(v["hbm_bytes_per_launch"] for v in out.values()) / 1e6, 1))
PY
ls $P | grep $TAG
