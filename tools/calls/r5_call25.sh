#!/bin/bash
# round 5, call 25: where between the round-4 tree and the current one did config 2 lose its ~1 %?  Trees of three intermediate commits
# (git archive + build in _ab_<commit>/) in alternation with both ends on one box
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5q
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # tag dir
  (cd $2 && timeout 600 python bench.py --config c2 --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/bis_$1.err | grep "^{" > $OUT/bis_$1.json)
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bis_$1.json"))
    print("c2 $1", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"])
except Exception as e:
    print("$1 FAILED", e); print(open("$OUT/bis_$1.err").read()[-800:])
PY
}
for rep in 1 2 3; do
  run r4_$rep $ROOT/_ab_r4
  for c in ${COMMITS:-2952df5 adf65bf dfbb4be}; do run ${c}_$rep $ROOT/_ab_$c; done
  run head_$rep $ROOT
done
