#!/bin/bash
# round 5, call 23: r4 tree vs current tree at config 2 - does the KV cache's capacity (bench.py sizes it for more steps now: 5504 rows instead of the
# 3712 of the round-4 bench in this command) explain the 1 %?  Alternating on one box
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5q
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # tag env...
  local tag=$1; shift
  (cd ${DIR:-$ROOT} && env "$@" timeout 600 python bench.py --config c2 --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/ms_$tag.err | grep "^{" > $OUT/ms_$tag.json)
  python - <<PY
import json
try:
    d=json.load(open("$OUT/ms_$tag.json"))
    print("c2 $tag", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], d["config"].get("kv_len_end"))
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/ms_$tag.err").read()[-1200:])
PY
}
for rep in 1 2 3 4; do
  DIR=$ROOT/_ab_r4 run r4_$rep A=1
  run r5_default_$rep A=1
  run r5_maxseq_as_r4_$rep LADE_BENCH_MAX_SEQ=${R4SEQ:-3652}
done
