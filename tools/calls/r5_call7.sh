#!/bin/bash
# round 5, call 7: the 160- and 224-row GEMM classes (a 132-row step no longer pads to 192): bit-identity / fp32-reference tests of the new
# work-group shapes, the engine suites that depend on the row classes, then c4 (whose steps with candidates fall into 129-256 rows) with the
# old classes and the new ones in alternation on one box: cold line, mid regime, hot regime
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ktile.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_parity_shapes.py -x -q -k "not dynamic" 2>&1 | tail -4
for rep in 1 2; do
  for v in r4 r5; do
    LADE_ROW_CLASSES=$v timeout 900 python bench.py --config c4 --steps 32 --warmup 8 --no-cpu-baseline --blocks 2 2> $OUT/rc_c4_${v}_$rep.err | grep "^{" > $OUT/rc_c4_${v}_$rep.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/rc_c4_${v}_$rep.json"))
    m,h,f=d["mid_regime"],d["hot_regime"],d["hot_regime_forced"]
    print("c4 classes=$v rep $rep cold", d["value"], d["ms_per_step"], "| mid S", m["step_compression"], "T", m["tokens_per_step_T"], "ms", m["ms_per_step"], "x plain", m["speedup_vs_plain"], "at 1.95:", m["speedup_at_published_S"]["1.95"],
          "| hot S", h["step_compression"], "T", h["tokens_per_step_T"], "ms", h["ms_per_step"], h["value"], "| forced T", f["tokens_per_step_T"], "ms", f["ms_per_step"], f["value"])
except Exception as e:
    print("c4 classes=$v $rep FAILED", e); print(open("$OUT/rc_c4_${v}_$rep.err").read()[-2500:])
PY
  done
done
