#!/bin/bash
# One-shot evidence capture on the GPU box:  gpurun --timeout 1500 -- 'bash tools/capture_profile.sh r01d'
# then, back in the container:                python tools/summarize_prof.py --tag r01d --stats ... (printed below)
# Counter passes are separate rocprofv3 runs with --kernel-trace only (no sys/hip/hsa tracing).
set -u
TAG=${1:-r01x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd "$ROOT"
if [ "${SKIP_TESTS:-0}" != 1 ]; then
    timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$OUT/pytest_gpu.log"
    tail -3 "$OUT/pytest_gpu.log"
fi
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 600 "$OUT/bench.json"
timeout 300 python tools/profile_ops.py --json "$OUT/ops.json" > "$OUT/ops.txt" 2>&1
timeout 300 python tools/profile_ops.py --dtype fp32 --batch 64 > "$OUT/ops_fp32_b64.txt" 2>&1
timeout 300 python tools/profile_ops.py --dtype fp32_split --batch 64 --json "$OUT/ops_split.json" > "$OUT/ops_split_b64.txt" 2>&1
# BASELINE configs[4] per-GPU shards (1280x1280, top-1000) and the configs[3] VGA bucket mix
timeout 300 python tools/profile_ops.py --size 1280 --topk 1000 --batch 4 --json "$OUT/ops_1280_b4.json" > "$OUT/ops_1280_b4.txt" 2>&1
timeout 300 python tools/profile_ops.py --size 1280 --topk 1000 --batch 32 --json "$OUT/ops_1280_b32.json" > "$OUT/ops_1280_b32.txt" 2>&1
timeout 300 python bench.py --size 1280 --topk 1000 --batch 4 --no-cpu-baseline --no-extras > "$OUT/bench_1280_b4.json" 2>> "$OUT/bench.err"
timeout 300 python bench.py --size 1280 --topk 1000 --batch 32 --no-cpu-baseline --no-extras > "$OUT/bench_1280_b32.json" 2>> "$OUT/bench.err"
timeout 300 python tools/vga_buckets_bench.py > "$OUT/vga_buckets.json" 2>> "$OUT/bench.err"
cd /tmp
# kernel durations with ONE context (what bench.py's roofline block times with HIP events), then the default two-context ring
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_k" -o k -- \
    python "$ROOT/bench.py" --depth 1 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-extras > "$OUT/prof_k.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_k2" -o k -- \
    python "$ROOT/bench.py" --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-extras > "$OUT/prof_k2.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_k1280" -o k -- \
    python "$ROOT/bench.py" --size 1280 --topk 1000 --batch 4 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-extras > "$OUT/prof_k1280.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_fetch" -o f -- \
    python "$ROOT/tools/profile_ops.py" --reps 3 > "$OUT/prof_fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/prof_write" -o w -- \
    python "$ROOT/tools/profile_ops.py" --reps 3 > "$OUT/prof_write.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES \
    --kernel-trace --output-format csv -d "$OUT/prof_inst" -o i -- \
    python "$ROOT/tools/profile_ops.py" --reps 3 > "$OUT/prof_inst.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM \
    --kernel-trace --output-format csv -d "$OUT/prof_sq" -o s -- \
    python "$ROOT/tools/profile_ops.py" --reps 3 > "$OUT/prof_sq.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS \
    --kernel-trace --output-format csv -d "$OUT/prof_lds" -o l -- \
    python "$ROOT/tools/profile_ops.py" --reps 3 > "$OUT/prof_lds.log" 2>&1
# tolerance mode (fp32_split): kernel durations, HBM traffic and the SQ view that shows what binds its kernels (LDS vs VALU vs MFMA)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_ksplit" -o k -- \
    python "$ROOT/bench.py" --dtype fp32_split --depth 1 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-extras > "$OUT/prof_ksplit.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM \
    --kernel-trace --output-format csv -d "$OUT/prof_sq_split" -o s -- \
    python "$ROOT/tools/profile_ops.py" --dtype fp32_split --reps 3 > "$OUT/prof_sq_split.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA \
    --kernel-trace --output-format csv -d "$OUT/prof_lds_split" -o l -- \
    python "$ROOT/tools/profile_ops.py" --dtype fp32_split --reps 3 > "$OUT/prof_lds_split.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/prof_fetch_split" -o f -- \
    python "$ROOT/tools/profile_ops.py" --dtype fp32_split --reps 3 > "$OUT/prof_fetch_split.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/prof_write_split" -o w -- \
    python "$ROOT/tools/profile_ops.py" --dtype fp32_split --reps 3 > "$OUT/prof_write_split.log" 2>&1
find "$OUT" -name '*.csv' | head -20
# keep the merge-back under 64 MiB: drop per-dispatch traces, keep stats + counters
find "$OUT" -name '*kernel_trace.csv' -size +8M -delete
du -sh "$OUT"
