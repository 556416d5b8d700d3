#!/bin/bash
# round 5, call 19: did the round move config 2?  The round-4 tree (git archive of the round-4 end state, built in _ab_r4/) against the
# current tree, the same bench command, alternating on one box
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5q
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2 3 4 5 6; do
  for v in r4 r5; do
    d=$ROOT; [ $v = r4 ] && d=$ROOT/_ab_r4
    (cd $d && timeout 600 python bench.py --config ${CFG:-c2} --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/r4r5_${v}_$rep.err | grep "^{" > $OUT/r4r5_${v}_$rep.json)
    python - <<PY
import json
try:
    d=json.load(open("$OUT/r4r5_${v}_$rep.json"))
    print("${CFG:-c2} tree $v rep $rep", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], [v.get("kernel") for k, v in d.get("projections", {}).items() if isinstance(v, dict) and "kernel" in v])
except Exception as e:
    print("$v $rep FAILED", e); print(open("$OUT/r4r5_${v}_$rep.err").read()[-1200:])
PY
  done
done
