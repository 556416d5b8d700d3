#!/bin/bash
# rocprofv3 counter passes of the attention launch pair at the bench's own launch shapes AND launch parameters (run on the GPU box from the repo root):
#   bash tools/attn_pmc.sh <out.json> [P] [bench_c2.json bench_c4.json bench_c5.json]
# One --pmc pass per counter (FETCH_SIZE and WRITE_SIZE cannot share a pass), --kernel-trace only; tools/pmc_summary.py turns each pair of
# passes into one entry {H, Hkv, d, T, P, n_splits, wg_rows, traffic_bytes} of <out.json> (read by bench.py for roofline.traffic).  The split count and
# the work-group rows of a shape are the ones the engine's in-step tuner chose in the bench line of that configuration
# (roofline.launch_parameters) when the line is given, else the default rule.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUTJ=$1
P=${2:-2190}
B2=${3:-}; B4=${4:-}; B5=${5:-}
RAW=${RAW_DIR:-$ROOT/gpurun_out/r5p/pmc}
mkdir -p $RAW
export TMPDIR=/tmp
pass() {   # label, counters, command...
    local label=$1 ctr=$2; shift 2
    rm -rf /tmp/prof_$label
    (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_$label -- "$@" > $RAW/$label.log 2>&1)
    local f=$(find /tmp/prof_$label -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $RAW/$label.csv || echo "no counter csv for $label"
}
params() {   # bench json, H, Hkv, T -> "n_splits wg_rows"
    python - "$1" "$2" "$3" "$4" "$P" <<PY
import json, sys
sys.path.insert(0, "$ROOT")
from lookaheaddecoding_amd import ops
path, H, Hkv, T, P = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
wg, mode = 0, 0
try:
    lp = json.load(open(path))["roofline"]["launch_parameters"]
    wg, mode = int(lp["wg_rows"]), int(lp["split_mode"])
except Exception:
    pass
ns = min(ops.choose_splits(H, H // Hkv, T, max(P + T, 1024), 256, allow_single=False, block_rows=0 if wg in (0, 128) else wg, mode=mode), 32)
print(ns, wg or 128)
PY
}
#        label  H  Hkv T   W  N  bench line
shapes=("c2    32 32  60  15 5 $B2" "c2g   32 32  120 15 5 $B2" "c4    40 40  120 20 7 $B4" "c5    64 8   60  15 5 $B5")
for sh in "${shapes[@]}"; do
    set -- $sh
    label=$1; H=$2; Hkv=$3; T=$4; W=$5; N=$6; BJ=${7:-none}
    read ns wg <<< "$(params $BJ $H $Hkv $T)"
    cmd="python $ROOT/tools/attn_bench.py --H $H --Hkv $Hkv --T $T --W $W --N $N --P $P --splits $ns --wg $wg --reps 40"
    echo "$label: n_splits $ns wg_rows $wg"
    pass ${label}_fetch "FETCH_SIZE" $cmd
    pass ${label}_write "WRITE_SIZE" $cmd
    pass ${label}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" $cmd
    python tools/pmc_summary.py $RAW/${label}_fetch.csv $RAW/${label}_write.csv $T $P $ns $OUTJ $H $Hkv 128 $wg
done
python tools/pmc_table.py $RAW/attn_pmc_table.json $(for l in c2 c2g c4 c5; do echo ${l}_mfma=$RAW/${l}_mfma.csv ${l}_fetch=$RAW/${l}_fetch.csv ${l}_write=$RAW/${l}_write.csv; done) --match attn_fwd_kernel --match attn_combine_kernel
