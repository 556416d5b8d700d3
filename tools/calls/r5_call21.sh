#!/bin/bash
# round 5, call 21: is the step sensitive to the size of the attention kernels' by-value argument struct?  The current library against a
# variant whose AttnK carries 128 more bytes (_ab_cur/liblade_hip.so), config 2, alternating on one box
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5q
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2 3 4; do
  for v in cur var; do
    lib=""; [ $v = var ] && lib=$ROOT/_ab_cur/liblade_hip.so
    LADE_HIP_LIB=$lib timeout 600 python bench.py --config c2 --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/ka_${v}_$rep.err | grep "^{" > $OUT/ka_${v}_$rep.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/ka_${v}_$rep.json"))
    print("c2 $v rep $rep", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"])
except Exception as e:
    print("$v $rep FAILED", e); print(open("$OUT/ka_${v}_$rep.err").read()[-1200:])
PY
  done
done
