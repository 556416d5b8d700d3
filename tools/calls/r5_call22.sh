#!/bin/bash
# round 5, call 22: the current tree is 1.2 % slower than the round-4 tree at config 2 on one box (call 19), the trace shows gaps at the head of
# the step's graph: does it come from what happens BEFORE the loop (decisions of every row class taken at construction: LADE_PREPARE, the attention
# coordinate of the in-step tuner: LADE_ATTN_TUNE)?  Alternating on one box
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5q
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # tag env...
  local tag=$1; shift
  (cd ${DIR:-$ROOT} && env "$@" timeout 600 python bench.py --config c2 --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/prep_$tag.err | grep "^{" > $OUT/prep_$tag.json)
  python - <<PY
import json
try:
    d=json.load(open("$OUT/prep_$tag.json"))
    print("c2 $tag", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/prep_$tag.err").read()[-1200:])
PY
}
for rep in 1 2 3; do
  DIR=$ROOT/_ab_r4 run r4_$rep A=1
  run r5_default_$rep A=1
  run r5_noprepare_$rep LADE_PREPARE=0
  run r5_noprepare_noattn_$rep LADE_PREPARE=0 LADE_ATTN_TUNE=0
done
