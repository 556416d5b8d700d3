#!/bin/bash
# round 5, call 24: r4 tree | current tree | current tree without the attention coordinate of the in-step tuner, config 2, six alternating rounds on one box
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5q
mkdir -p $OUT
export TMPDIR=/tmp
run() {  # tag env...
  local tag=$1; shift
  (cd ${DIR:-$ROOT} && env "$@" timeout 600 python bench.py --config c2 --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/at_$tag.err | grep "^{" > $OUT/at_$tag.json)
  python - <<PY
import json
try:
    d=json.load(open("$OUT/at_$tag.json"))
    a=(d.get("projections",{}).get("in_step_tuning",{}).get("attn") or {}).get("in_step_choice")
    print("c2 $tag", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], a, [v.get("kernel") for k, v in d.get("projections", {}).items() if isinstance(v, dict) and "kernel" in v])
except Exception as e:
    print("$tag FAILED", e); print(open("$OUT/at_$tag.err").read()[-1200:])
PY
}
for rep in 1 2 3 4; do
  DIR=$ROOT/_ab_r4 run r4_$rep A=1
  run r5_default_$rep A=1
  run r5_yield_rarely_$rep LADE_POLL_READS_PER_YIELD=1000000
done
