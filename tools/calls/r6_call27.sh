#!/bin/bash
# round 6, call 27: the attention kernel's output / partial rows stored write-through (sc1; -DLADE_PO_WT=1, one library per arm) against plain stores, in the step (c2 / c4)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6aa
mkdir -p $OUT
L=$PWD/lookaheaddecoding_amd
for rep in 1 2 3; do
  for arm in base powt; do
    lib=$L/liblade_hip_$arm.so; [ $arm = base ] && lib=$L/liblade_hip.so
    for c in c2 c4; do
      LADE_HIP_LIB=$lib timeout 1200 python bench.py --config $c --no-cpu-baseline --no-generate --blocks 3 2> $OUT/${c}_${arm}_$rep.err | grep "^{" > $OUT/${c}_${arm}_$rep.json
      python - <<PY
import json
try:
    d=json.load(open("$OUT/${c}_${arm}_$rep.json"))
    r=d["roofline"]
    print("$c $arm rep $rep cold", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "| pair us", r["launch_us"], "frac", r["frac"], "| plain", d["plain_decode"]["ms_per_token"], "| mid ms", d["mid_regime"]["ms_per_step"], "| hot ms", d["hot_regime"]["ms_per_step"])
except Exception as e:
    print("$c $arm $rep FAILED", e); print(open("$OUT/${c}_${arm}_$rep.err").read()[-600:])
PY
    done
  done
done | tee $OUT/attn_output_store_policy_ab.txt
