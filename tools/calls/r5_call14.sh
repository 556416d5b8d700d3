#!/bin/bash
# round 5, call 14: GEMM K loop with ONE barrier per K tile (the next tile is requested behind the barrier that frees its slot, before the
# MFMA phase) = _ab_cur/liblade_hip.so, against the kept two-barrier loop: bit-identity tests on the variant, the four projections in
# isolation (tools/gemm_ingest_probe.py), then c4 and c2 alternating on one box
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5q
mkdir -p $OUT
export TMPDIR=/tmp
VAR=$ROOT/_ab_cur/liblade_hip.so
LADE_HIP_LIB=$VAR timeout 900 python -m pytest tests/test_gpu_ktile.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -3
for rep in 1 2; do
  for v in cur var; do
    lib=""; [ $v = var ] && lib=$VAR
    echo "$v: $(LADE_HIP_LIB=$lib MODEL=13b M=120 python tools/gemm_ingest_probe.py 2>&1 | tail -1)"
    echo "$v: $(LADE_HIP_LIB=$lib MODEL=7b M=60 python tools/gemm_ingest_probe.py 2>&1 | tail -1)"
  done
done
run() {   # cfg variant rep
  local lib=""; [ $2 = var ] && lib=$VAR
  LADE_HIP_LIB=$lib timeout 900 python bench.py --config $1 --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/e1_$1_$2_$3.err | grep "^{" > $OUT/e1_$1_$2_$3.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/e1_$1_$2_$3.json"))
    print("$1 $2 $3", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], [v.get("kernel") for k, v in d.get("projections", {}).items() if isinstance(v, dict) and "kernel" in v])
except Exception as e:
    print("$1 $2 $3 FAILED", e); print(open("$OUT/e1_$1_$2_$3.err").read()[-1500:])
PY
}
for rep in 1 2; do for v in cur var; do run c4 $v $rep; done; done
for rep in 1 2; do for v in cur var; do run c2 $v $rep; done; done
