#!/bin/bash
# Quick counter capture on the GPU box for kernels matching a pattern:
#   gpurun --timeout 900 -- 'bash tools/pmc_quick.sh <tag> <grep pattern> [profile_ops args]'
# Three separate rocprofv3 --pmc passes (instruction counts, SQ cycles, LDS) over tools/profile_ops.py, --kernel-trace only.
set -u
TAG=${1:-q}; PAT=${2:-cf::}; shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
run() {  # name, counters...
    local n=$1; shift
    timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$n" -o p -- python "$ROOT/tools/profile_ops.py" --reps 2 "${EXTRA[@]}" > "$OUT/$n.log" 2>&1
    local f=$(find "$OUT/$n" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python "$ROOT/tools/pmc_summary.py" "$f" | grep -E "kernel|$PAT" > "$OUT/$n.txt"
    find "$OUT/$n" -name '*.csv' -size +4M -delete
    cat "$OUT/$n.txt"
}
EXTRA=("$@")
run inst SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
run lds SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/k" -o k -- python "$ROOT/tools/profile_ops.py" --reps 3 "${EXTRA[@]}" > "$OUT/k.log" 2>&1
f=$(find "$OUT/k" -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && grep -E "Name|$PAT" "$f" | cut -c1-200 | tee "$OUT/k.txt"
find "$OUT" -name '*kernel_trace.csv' -size +4M -delete
