#!/bin/bash
# Counters of the STANDALONE depthwise kernels (unfused path): HBM traffic per launch next to the algorithmic bytes, SQ busy / wait shares.
#   gpurun --timeout 900 -- 'bash tools/dw_pmc.sh <tag>'
# Separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ set) over tools/profile_ops.py --no-fuse, --kernel-trace only.
set -u
TAG=${1:-dwpmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
python "$ROOT/tools/profile_ops.py" --no-fuse --reps 5 2>/dev/null | grep -E "layer|\.dw" > "$OUT/time.txt"
cat "$OUT/time.txt"
run() {
    local n=$1; shift
    timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$n" -o p -- python "$ROOT/tools/profile_ops.py" --no-fuse --reps 2 > "$OUT/$n.log" 2>&1
    local f=$(find "$OUT/$n" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python "$ROOT/tools/pmc_summary.py" "$f" | grep -E "kernel|dw_" > "$OUT/$n.txt"
    find "$OUT/$n" -name '*.csv' -size +4M -delete
    cat "$OUT/$n.txt"
}
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum
