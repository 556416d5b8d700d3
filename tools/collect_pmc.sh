#!/bin/bash
# rocprofv3 counter passes behind profiles/r2_*  (run on the GPU box from the repo root: bash tools/collect_pmc.sh)
# One --pmc pass per counter group, --kernel-trace only (never with the hip/hsa trace domains).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/pmc
mkdir -p $OUT
export TMPDIR=/tmp
run() {   # label, counters, command...
    local label=$1 ctr=$2; shift 2
    rm -rf /tmp/prof_$label
    (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_$label -- "$@" > $OUT/$label.log 2>&1)
    local f=$(find /tmp/prof_$label -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $OUT/$label.csv || echo "no counter csv for $label"
}
A7="python $ROOT/tools/attn_bench.py --T 60 --P 2016 --splits 0 --reps 40"
A7B="python $ROOT/tools/attn_bench.py --T 120 --P 2016 --splits 0 --reps 40"
A70="python $ROOT/tools/attn_bench.py --lp-rank --H 64 --Hkv 8 --P 2016 --splits 0 --reps 40"
G128="env M=128 python $ROOT/tools/gemm_flags.py"
G60="env M=60 python $ROOT/tools/gemm_flags.py"
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
LD="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
for w in A7 A7B A70 G128 G60; do
    cmd=${!w}
    run ${w}_mfma "$MF" $cmd
    run ${w}_lds "$LD" $cmd
    run ${w}_fetch "FETCH_SIZE" $cmd
    run ${w}_write "WRITE_SIZE" $cmd
done
ls -la $OUT
