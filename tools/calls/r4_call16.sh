#!/bin/bash
# round 4, call 16: GEMM with separate weight / activation rings fed by separate waves - bit-identity tests, then c2 and c4 against the
# previous build (_ab_cur/ = the unified ring), alternating on one box
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ktile.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -6
for cfg in c2 c4; do
  for rep in 1 2; do
    for v in cur new; do
      d=$ROOT; [ $v = cur ] && d=$ROOT/_ab_cur
      (cd $d && LADE_TUNE_VERBOSE=1 timeout 600 python bench.py --config $cfg --steps 32 --warmup 8 --no-cpu-baseline --blocks 3 2> $OUT/sr_${cfg}_${v}_$rep.err | grep "^{" > $OUT/sr_${cfg}_${v}_$rep.json)
      python - <<PY
import json
try:
    d=json.load(open("$OUT/sr_${cfg}_${v}_$rep.json"))
    print("$cfg $v $rep", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "plain", d["plain_decode"]["ms_per_token"], "hot", d["hot_regime"]["value"], [v.get("kernel") for k, v in d.get("projections", {}).items() if isinstance(v, dict) and "kernel" in v])
except Exception as e:
    print("$cfg $v $rep FAILED", e); print(open("$OUT/sr_${cfg}_${v}_$rep.err").read()[-1500:])
PY
    done
  done
done
