#!/bin/bash
# round 6, call 25: the GEMMs' fp32 split-K partial STORES non-temporal (LADE_DEBUG=gemm_dbg=64) / write-through (gemm_dbg=8) against plain stores, IN THE STEP (c2 / c4,
# shipped table).  The switch sits in the epilogue's store loop, outside the K loop, and both arms run the same branches (round 2 measured these isolated: faster GEMM, slower consumer)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6y
mkdir -p $OUT
for rep in 1 2; do
  for arm in plain nt wt; do
    dbg=""; [ $arm = nt ] && dbg="gemm_dbg=64"; [ $arm = wt ] && dbg="gemm_dbg=8"
    for c in c2 c4; do
      LADE_DEBUG=$dbg timeout 1200 python bench.py --config $c --no-cpu-baseline --no-generate --blocks 3 2> $OUT/${c}_${arm}_$rep.err | grep "^{" > $OUT/${c}_${arm}_$rep.json
      python - <<PY
import json
try:
    d=json.load(open("$OUT/${c}_${arm}_$rep.json"))
    print("$c $arm rep $rep cold", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "| plain", d["plain_decode"]["ms_per_token"], "| mid ms", d["mid_regime"]["ms_per_step"], "T", d["mid_regime"]["tokens_per_step_T"], "| hot ms", d["hot_regime"]["ms_per_step"])
except Exception as e:
    print("$c $arm $rep FAILED", e); print(open("$OUT/${c}_${arm}_$rep.err").read()[-600:])
PY
    done
  done
done | tee $OUT/partial_store_policy_ab.txt
