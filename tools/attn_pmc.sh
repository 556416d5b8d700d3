#!/bin/bash
# rocprofv3 counter passes of the attention launch pair at the bench's own launch shapes (run on the GPU box from the repo root):
#   bash tools/attn_pmc.sh <out.json> [P]
# One --pmc pass per counter (FETCH_SIZE and WRITE_SIZE cannot share a pass), --kernel-trace only; tools/pmc_summary.py turns each pair of
# passes into one entry {H, Hkv, d, T, P, n_splits, traffic_bytes} of <out.json> (read by bench.py for roofline.traffic).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUTJ=$1
P=${2:-2190}
RAW=${RAW_DIR:-$ROOT/gpurun_out/r4p/pmc}
mkdir -p $RAW
export TMPDIR=/tmp
pass() {   # label, counters, command...
    local label=$1 ctr=$2; shift 2
    rm -rf /tmp/prof_$label
    (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_$label -- "$@" > $RAW/$label.log 2>&1)
    local f=$(find /tmp/prof_$label -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $RAW/$label.csv || echo "no counter csv for $label"
}
#        label  H  Hkv T   W  N
shapes=("c2    32 32  60  15 5" "c2g   32 32  120 15 5" "c4    40 40  120 20 7" "c5    64 8   60  15 5")
for sh in "${shapes[@]}"; do
    set -- $sh
    label=$1; H=$2; Hkv=$3; T=$4; W=$5; N=$6
    cmd="python $ROOT/tools/attn_bench.py --H $H --Hkv $Hkv --T $T --W $W --N $N --P $P --splits 0 --reps 40"
    ns=$(python - <<PY
import sys; sys.path.insert(0, "$ROOT")
from lookaheaddecoding_amd import ops
print(min(ops.choose_splits($H, $H // $Hkv, $T, max($P + $T, 1024), 256, allow_single=False), 32))
PY
)
    pass ${label}_fetch "FETCH_SIZE" $cmd
    pass ${label}_write "WRITE_SIZE" $cmd
    pass ${label}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $cmd
    python tools/pmc_summary.py $RAW/${label}_fetch.csv $RAW/${label}_write.csv $T $P $ns $OUTJ $H $Hkv 128
done
python tools/pmc_table.py $RAW/attn_pmc_table.json $(for l in c2 c2g c4 c5; do echo ${l}_mfma=$RAW/${l}_mfma.csv ${l}_fetch=$RAW/${l}_fetch.csv ${l}_write=$RAW/${l}_write.csv; done) --match attn_fwd_kernel --match attn_combine_kernel
