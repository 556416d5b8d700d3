#!/bin/bash
# round 5, call 18: the sampling step's draw without torch.multinomial's input checks (sampling.multinomial_one) + pinned staging of the
# forced tokens: identity test, sampling suites, then c3 with LADE_DRAW_TORCH=1 (torch.multinomial) and without, alternating on one box
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k multinomial 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_examples.py -x -q -k "sampl" 2>&1 | tail -2
for rep in 1 2 3; do
  for v in 1 0; do
    LADE_DRAW_TORCH=$v timeout 900 python bench.py --config c3 --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/draw_${v}_$rep.err | grep "^{" > $OUT/draw_${v}_$rep.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/draw_${v}_$rep.json"))
    print("c3 LADE_DRAW_TORCH=$v rep $rep", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"])
except Exception as e:
    print("c3 $v $rep FAILED", e)
PY
  done
done
