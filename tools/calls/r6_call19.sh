#!/bin/bash
# round 6, call 19: cache-policy bits as COMPILE-time constants, one library per arm (tools/build_variant.sh): base (weights nt, K / V default = rounds 2-5), kv18 (K / V
# nt + sc1), w18 (weights nt + sc1), both18 - c2 and c4 bench lines alternating on one box, the shipped decision table in every arm.  (Calls 16-18 compared run-time
# switches: the extra case in the GEMM's DMA issue loop cost the step ~5 % by itself and fell unevenly on the arms.)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6t
mkdir -p $OUT
L=$PWD/lookaheaddecoding_amd
for rep in 1 2; do
  for arm in base kv18 w18 both18; do
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
done | tee $OUT/cache_policy_variants_ab.txt
