#!/bin/bash
# round 4, call 5: (a) the staged post-step + the refine-hook fix: GPU tests of the integer kernels, end-to-end traces, full-size
# successor models; (b) A/B on one box, alternating: round-3 tree built with a 4-stage ring (_ab_old/) vs this tree with the isolated
# table vs this tree with the in-step refinement
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_ktile.py -x -q 2>&1 | tail -8
B="bench.py --steps 32 --warmup 8 --no-cpu-baseline --blocks 3"
for rep in 1 2; do
  for v in old iso step; do
    case $v in
      old) (cd _ab_old && timeout 400 python $B 2> $ROOT/$OUT/abx_${v}_$rep.err | grep "^{" > $ROOT/$OUT/abx_${v}_$rep.json) ;;
      iso) LADE_TUNE_STEP=0 timeout 400 python $B 2> $OUT/abx_${v}_$rep.err | grep "^{" > $OUT/abx_${v}_$rep.json ;;
      step) LADE_TUNE_STEP=1 timeout 400 python $B 2> $OUT/abx_${v}_$rep.err | grep "^{" > $OUT/abx_${v}_$rep.json ;;
    esac
    python - <<PY
import json
d=json.load(open("$OUT/abx_${v}_$rep.json"))
print("$v $rep", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "plain", d["plain_decode"]["ms_per_token"], "hot", d["hot_regime"]["value"], "pair", d["roofline"]["launch_us"], [v.get("kernel") for k, v in d.get("projections", {}).items() if isinstance(v, dict) and "kernel" in v])
PY
  done
done
