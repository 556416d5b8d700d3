#!/bin/bash
# round 4, call 4: which change breaks the successor-model end-to-end test (in-step tuning on / off); static vs launch-parameter ring A/B
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
echo "== LADE_TUNE_STEP=0"; LADE_TUNE_STEP=0 timeout 300 python tools/calls/r4_dbg_succ.py 2>&1 | grep -v amdgpu.ids | cut -c1-700
echo "== LADE_TUNE_STEP=1"; LADE_TUNE_STEP=1 timeout 300 python tools/calls/r4_dbg_succ.py 2>&1 | grep -v amdgpu.ids | cut -c1-700
B="python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3"
for rep in 1 2; do
  for v in dyn st4; do
    lib=""; [ $v = st4 ] && lib="$PWD/lookaheaddecoding_amd/liblade_hip_st4.so"
    LADE_TUNE_STEP=0 LADE_HIP_LIB=$lib timeout 300 $B 2> $OUT/ring_${v}_$rep.err | grep "^{" > $OUT/ring_${v}_$rep.json
    python - <<PY
import json
d=json.load(open("$OUT/ring_${v}_$rep.json"))
print("$v $rep", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], [ (k, v.get("kernel")) for k, v in d.get("projections", {}).items() if isinstance(v, dict) and "kernel" in v])
PY
  done
done
