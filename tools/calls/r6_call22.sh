#!/bin/bash
# round 6, call 22: three more cache-policy constants, one library per arm, in the step (c2 / c4, shipped table): a16 = the GEMMs' ACTIVATION tile past the CU's L1 (sc1);
# kv2 = K / V stream non-temporal WITHOUT sc1; kv19 = nt + sc0 + sc1; base = the round's final library (K / V nt + sc1, weights nt)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6v
mkdir -p $OUT
L=$PWD/lookaheaddecoding_amd
for rep in 1 2; do
  for arm in base a16 kv2 kv19; do
    lib=$L/liblade_hip_$arm.so; [ $arm = base ] && lib=$L/liblade_hip.so
    for c in c2 c4; do
      LADE_HIP_LIB=$lib timeout 1200 python bench.py --config $c --no-cpu-baseline --no-generate --blocks 3 2> $OUT/${c}_${arm}_$rep.err | grep "^{" > $OUT/${c}_${arm}_$rep.json
      python - <<PY
import json
try:
    d=json.load(open("$OUT/${c}_${arm}_$rep.json"))
    r=d["roofline"]
    print("$c $arm rep $rep cold", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "| pair us", r["launch_us"], "frac", r["frac"], "| plain", d["plain_decode"]["ms_per_token"], "| mid ms", d["mid_regime"]["ms_per_step"], "T", d["mid_regime"]["tokens_per_step_T"], "| hot ms", d["hot_regime"]["ms_per_step"], "| prefill", d["prefill"]["tokens_per_s"])
except Exception as e:
    print("$c $arm $rep FAILED", e); print(open("$OUT/${c}_${arm}_$rep.err").read()[-600:])
PY
    done
  done
done | tee $OUT/cache_policy_variants_ab2.txt
