#!/bin/bash
# the GQA attention launch (config 5: H = 64, Hkv = 8, T = 60, P ~ 2190) judged by its HBM TRAFFIC (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE
# over attn_fwd + attn_combine, separate --pmc passes) next to its time, for work-group rows x split counts:  bash tools/attn_gqa_sweep.sh <out.txt>
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$1
P=${P:-2190}
RAW=/tmp/gqa_sweep
mkdir -p $RAW
export TMPDIR=/tmp
pass() {
    local label=$1 ctr=$2; shift 2
    rm -rf /tmp/prof_$label
    (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_$label -- "$@" > $RAW/$label.log 2>&1)
    local f=$(find /tmp/prof_$label -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $RAW/$label.csv || echo "no counter csv for $label"
}
echo "# config 5 attention pair (H=64 Hkv=8 d=128 T=60 P=$P bf16): us per pair (400 launches over rotating caches) | HBM bytes per pair / algorithmic" > $OUT
for combo in "128 6" "128 4" "128 3" "128 2" "64 6" "64 4" "64 3" "64 2" "32 3" "32 2"; do
    set -- $combo; wg=$1; ns=$2
    t=$(python tools/attn_bench.py --H 64 --Hkv 8 --T 60 --P $P --splits $ns --wg $wg --reps 400 2>/dev/null | grep "splits=" | head -1)
    cmd="python $ROOT/tools/attn_bench.py --H 64 --Hkv 8 --T 60 --P $P --splits $ns --wg $wg --reps 40"
    pass g_${wg}_${ns}_fetch FETCH_SIZE $cmd
    pass g_${wg}_${ns}_write WRITE_SIZE $cmd
    tr=$(python tools/pmc_summary.py $RAW/g_${wg}_${ns}_fetch.csv $RAW/g_${wg}_${ns}_write.csv 60 $P $ns $RAW/gqa.json 64 8 128 $wg | python -c "import json,sys; e=json.loads(sys.stdin.read()); print(e['traffic_bytes'], e['traffic_over_algorithmic'], e['fetch_kib_raw'], e['write_kib_raw'])")
    echo "wg=$wg splits=$ns | $t | traffic $tr" | tee -a $OUT
done
