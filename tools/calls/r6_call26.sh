#!/bin/bash
# round 6, call 26: write-through split-K partial stores as the default against plain stores (LADE_DEBUG=gemm_dbg=8), second box: c2 / c4 / c3 alternating; then the GPU suite
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6z
mkdir -p $OUT
for rep in 1 2 3; do
  for arm in wt plain; do
    dbg=""; [ $arm = plain ] && dbg="gemm_dbg=8"
    for c in c2 c4; do
      LADE_DEBUG=$dbg timeout 1200 python bench.py --config $c --no-cpu-baseline --no-generate --blocks 3 2> $OUT/${c}_${arm}_$rep.err | grep "^{" > $OUT/${c}_${arm}_$rep.json
      python - <<PY
import json
try:
    d=json.load(open("$OUT/${c}_${arm}_$rep.json"))
    print("$c $arm rep $rep cold", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "| plain", d["plain_decode"]["ms_per_token"], "| mid ms", d["mid_regime"]["ms_per_step"], "T", d["mid_regime"]["tokens_per_step_T"], "| hot ms", d["hot_regime"]["ms_per_step"], "| prefill", d["prefill"]["tokens_per_s"])
except Exception as e:
    print("$c $arm $rep FAILED", e); print(open("$OUT/${c}_${arm}_$rep.err").read()[-600:])
PY
    done
  done
done | tee $OUT/partial_store_policy_ab2.txt
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tee $OUT/pytest_gpu.txt
