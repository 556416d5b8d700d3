#!/bin/bash
# round 6, call 17: cache-policy bits on the attention kernel's K / V stream (LADE_DEBUG=attn_dbg=128: sc1; 256: nt + sc1) against the default policy, IN THE STEP
# (c2 / c4 bench lines alternating on one box; the shipped decision table in every arm)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6r
mkdir -p $OUT
for rep in 1 2; do
  for arm in default sc1 ntsc1; do
    dbg=""; [ $arm = sc1 ] && dbg="attn_dbg=128"; [ $arm = ntsc1 ] && dbg="attn_dbg=256"
    for c in c2 c4; do
      LADE_DEBUG=$dbg timeout 1200 python bench.py --config $c --no-cpu-baseline --no-generate --blocks 3 2> $OUT/${c}_${arm}_$rep.err | grep "^{" > $OUT/${c}_${arm}_$rep.json
      python - <<PY
import json
try:
    d=json.load(open("$OUT/${c}_${arm}_$rep.json"))
    r=d["roofline"]
    print("$c $arm rep $rep cold", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "| pair us", r["launch_us"], "isolated", r["launch_us_isolated"], "frac", r["frac"], "| mid ms", d["mid_regime"]["ms_per_step"], "T", d["mid_regime"]["tokens_per_step_T"])
except Exception as e:
    print("$c $arm $rep FAILED", e); print(open("$OUT/${c}_${arm}_$rep.err").read()[-600:])
PY
    done
  done
done | tee $OUT/attn_kv_cache_policy_ab.txt
