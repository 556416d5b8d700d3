#!/bin/bash
# round 4, call 1: baseline bench line; LDS ring of 4 stages (build variant) A/B in alternation; wide-GEMM probe (> 256 rows on the
# hand-written kernel incl. the new 256 x 256 tile vs hipBLASLt); first f16 bench line
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3"
for rep in 1 2; do
  for v in base ns4; do
    lib=""; [ $v = ns4 ] && lib="$PWD/lookaheaddecoding_amd/liblade_hip_ns4.so"
    LADE_HIP_LIB=$lib timeout 300 $B 2> $OUT/ab_${v}_$rep.err | grep "^{" > $OUT/ab_${v}_$rep.json
    python - <<PY
import json
d=json.load(open("$OUT/ab_${v}_$rep.json"))
print("$v $rep", d["value"], d["ms_per_step"], d.get("spread",{}).get("block_ms"))
PY
  done
done
timeout 400 python tools/gemm_wide_probe.py 7b 512 847 2304 > $OUT/gemm_wide_7b.txt 2>&1; tail -20 $OUT/gemm_wide_7b.txt
timeout 300 python tools/gemm_wide_probe.py 70b 512 2304 > $OUT/gemm_wide_70b.txt 2>&1; tail -12 $OUT/gemm_wide_70b.txt
timeout 300 python tools/gemm_wide_probe.py 13b 240 > $OUT/gemm_wide_13b.txt 2>&1; tail -6 $OUT/gemm_wide_13b.txt
timeout 300 python bench.py --dtype f16 --steps 32 --warmup 8 --no-cpu-baseline --blocks 3 2> $OUT/bench_c2_f16.err | grep "^{" > $OUT/bench_c2_f16.json
cut -c1-300 $OUT/bench_c2_f16.json; tail -3 $OUT/bench_c2_f16.err
