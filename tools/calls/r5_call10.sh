#!/bin/bash
# round 5, call 10: counters of the step's four GEMM launches at the 128-row class (13B, 120 rows: the class the review singled out, 34 % above its bare weight
# stream) next to the 64-row class (7B, 60 rows): where do the waves wait?  Separate rocprofv3 --pmc passes, --kernel-trace only, of tools/gemm_ingest_probe.py
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5p/gemm_pmc
mkdir -p $OUT
export TMPDIR=/tmp
run() {   # label, counters, env..., command
    local label=$1 ctr=$2; shift 2
    rm -rf /tmp/prof_$label
    (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_$label -- "$@" > $OUT/$label.log 2>&1)
    local f=$(find /tmp/prof_$label -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $OUT/$label.csv || echo "no counter csv for $label"
}
WT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
LD="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16"
for w in 13b:120 7b:60; do
    model=${w%%:*}; m=${w##*:}
    run g${model}_${m}_wait "$WT" env MODEL=$model M=$m python $ROOT/tools/gemm_ingest_probe.py
    run g${model}_${m}_lds "$LD" env MODEL=$model M=$m python $ROOT/tools/gemm_ingest_probe.py
    run g${model}_${m}_fetch "FETCH_SIZE" env MODEL=$model M=$m python $ROOT/tools/gemm_ingest_probe.py
    run g${model}_${m}_write "WRITE_SIZE" env MODEL=$model M=$m python $ROOT/tools/gemm_ingest_probe.py
    tail -1 $OUT/g${model}_${m}_wait.log | cut -c1-400
done
python tools/pmc_table.py $OUT/gemm_pmc_table.json $(for l in g13b_120 g7b_60; do for k in wait lds fetch write; do echo ${l}_${k}=$OUT/${l}_${k}.csv; done; done) --match gemm_skinny_kernel
python - <<PY
import json
d=json.load(open("$OUT/gemm_pmc_table.json"))
for label in sorted(d):
    for k,v in d[label].items():
        print(label, k[-60:], {a:(round(b,1) if isinstance(b,float) else b) for a,b in v.items()})
PY
