#!/bin/bash
# round 4, call 3: the in-step GEMM refinement (decisions re-taken in hipGraphs of a real 8-layer forward) against the isolated table
# with the default 4-stage ring, alternating on one box
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 32 --warmup 8 --no-cpu-baseline --blocks 3"
for rep in 1 2; do
  for v in iso step; do
    ts=1; [ $v = iso ] && ts=0
    LADE_TUNE_STEP=$ts LADE_TUNE_VERBOSE=1 timeout 400 $B 2> $OUT/ts_${v}_$rep.err | grep "^{" > $OUT/ts_${v}_$rep.json
    python - <<PY
import json
d=json.load(open("$OUT/ts_${v}_$rep.json"))
print("$v $rep", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "plain", d["plain_decode"]["ms_per_token"], "hot", d["hot_regime"]["value"] if d.get("hot_regime") else None, "pair", d["roofline"]["launch_us"])
PY
  done
done
grep "tune-step" $OUT/ts_step_1.err | cut -c1-400
timeout 500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -5
