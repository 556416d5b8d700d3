#!/bin/bash
# round 5, call 12: add + RMSNorm with one 16-byte chunk per thread up to 1024 threads (hidden 5120 / 8192 ran two chunks per thread: the
# row paid its memory latency twice) - kernel tests, then c4 / c5 / c2 against the previous build (_ab_cur/liblade_hip.so via LADE_HIP_LIB),
# alternating on one box
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5q
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ktile.py -x -q 2>&1 | tail -4
run() {   # cfg variant rep
  local lib=""; [ $2 = old ] && lib=$ROOT/_ab_cur/liblade_hip.so
  LADE_HIP_LIB=$lib timeout 900 python bench.py --config $1 --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/norm_$1_$2_$3.err | grep "^{" > $OUT/norm_$1_$2_$3.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/norm_$1_$2_$3.json"))
    print("$1 $2 $3", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"])
except Exception as e:
    print("$1 $2 $3 FAILED", e); print(open("$OUT/norm_$1_$2_$3.err").read()[-1500:])
PY
}
for rep in 1 2; do for v in old new; do run c4 $v $rep; done; done
for v in old new; do run c5 $v 1; done
for v in old new; do run c2 $v 1; done
