#!/bin/bash
# round 6, call 16: nt + sc1 on the GEMMs' weight stream (new default) against nt only (LADE_DEBUG=gemm_dbg=256 = rounds 2-5) in the real step: c2 and c4 bench lines
# alternating on one box (in-process tuning in both arms: LADE_TUNE_FILE=off - the shipped table was tuned under the old policy), then the rows curves
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6q
mkdir -p $OUT
for rep in 1 2; do
  for arm in sc1 nt; do
    dbg=""; [ $arm = nt ] && dbg="gemm_dbg=256"
    for c in c2 c4; do
      LADE_TUNE_FILE=off LADE_DEBUG=$dbg timeout 1200 python bench.py --config $c --no-cpu-baseline --no-generate --blocks 3 2> $OUT/${c}_${arm}_$rep.err | grep "^{" > $OUT/${c}_${arm}_$rep.json
      python - <<PY
import json
try:
    d=json.load(open("$OUT/${c}_${arm}_$rep.json"))
    m,h=d["mid_regime"],d["hot_regime"]
    print("$c $arm rep $rep cold", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "| plain", d["plain_decode"]["ms_per_token"], "| mid S", m["step_compression"], "T", m["tokens_per_step_T"], "ms", m["ms_per_step"], "at S:", m["speedup_at_published_S"], "| hot T", h["tokens_per_step_T"], "ms", h["ms_per_step"], "| prefill", d["prefill"]["tokens_per_s"])
except Exception as e:
    print("$c $arm $rep FAILED", e); print(open("$OUT/${c}_${arm}_$rep.err").read()[-600:])
PY
    done
  done
done | tee $OUT/weight_policy_step_ab.txt
{
LADE_TUNE_FILE=off timeout 900 python tools/rows_curve.py 7b 1 60 92 120 150 180 240 2>&1 | grep -v amdgpu.ids | cut -c1-110
LADE_TUNE_FILE=off LADE_DEBUG=gemm_dbg=256 timeout 900 python tools/rows_curve.py 7b 1 60 92 120 150 180 240 2>&1 | grep -v amdgpu.ids | cut -c1-110
LADE_TUNE_FILE=off timeout 900 python tools/rows_curve.py 13b 1 60 92 120 150 180 240 2>&1 | grep -v amdgpu.ids | cut -c1-110
LADE_TUNE_FILE=off LADE_DEBUG=gemm_dbg=256 timeout 900 python tools/rows_curve.py 13b 1 60 92 120 150 180 240 2>&1 | grep -v amdgpu.ids | cut -c1-110
} | tee $OUT/rows_curve_weight_policy.txt
