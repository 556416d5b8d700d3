#!/bin/bash
# round 4, call 24: RoPE + KV append with the cache length and the token position read through vector loads and first used where they
# are needed (the partial loads no longer wait for two scalar round trips).  Kernel tests, then c2 against the previous library
# (liblade_hip_prev.so) in alternation on one box, then the glue microbenchmark of both
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -4
for rep in 1 2 3; do
  for v in prev new; do
    L=$ROOT/lookaheaddecoding_amd/liblade_hip.so; [ $v = prev ] && L=$ROOT/lookaheaddecoding_amd/liblade_hip_prev.so
    LADE_HIP_LIB=$L timeout 600 python bench.py --config c2 --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/rp_c2_${v}_$rep.err | grep "^{" > $OUT/rp_c2_${v}_$rep.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/rp_c2_${v}_$rep.json"))
    print("c2 $v rep $rep", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"])
except Exception as e:
    print("c2 $v $rep FAILED", e); print(open("$OUT/rp_c2_${v}_$rep.err").read()[-1500:])
PY
  done
done
for v in prev new; do
  L=$ROOT/lookaheaddecoding_amd/liblade_hip.so; [ $v = prev ] && L=$ROOT/lookaheaddecoding_amd/liblade_hip_prev.so
  echo "glue_bench $v"; LADE_HIP_LIB=$L timeout 300 python tools/glue_bench.py 2>&1 | grep -i "rope" | head -6
done
