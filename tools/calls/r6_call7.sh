#!/bin/bash
# round 6, call 7: (1) the tests of the measured-and-dropped kernels on the EXPERIMENTAL build (tools/build_variant.sh exp "" EXPERIMENTAL=1: fused-RoPE attention
# forms, RA GEMM, ping-pong K loop) - the default build skips them; (2) config 5 with the GQA attention launch at 64-row work-groups x 4 splits against the
# default 128 x 6 (profiles/r6_attn_gqa_sweep.txt: 1.85 x instead of 2.25 x the algorithmic bytes at equal isolated time), alternating on one box
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6g
mkdir -p $OUT
LADE_HIP_LIB=$PWD/lookaheaddecoding_amd/liblade_hip_exp.so timeout 1200 python -m pytest tests/test_gpu_fused_rope.py tests/test_gpu_ktile.py -m gpu -q 2>&1 | tail -6 | tee $OUT/pytest_experimental_build.txt
for rep in 1 2; do
  for arm in default gqa64x4; do
    dbg=""; [ $arm = gqa64x4 ] && dbg="attn_shape=64,attn_splits=4"
    LADE_DEBUG=$dbg timeout 1500 python bench.py --config c5 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/c5_${arm}_$rep.err | grep "^{" > $OUT/c5_${arm}_$rep.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/c5_${arm}_$rep.json"))
    print("c5 $arm rep $rep", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "pair us", d["roofline"]["launch_us"], d["roofline"]["launch_us_source"][:40], "splits", d["roofline"]["launch_parameters"]["n_splits"])
except Exception as e:
    print("$arm $rep FAILED", e); print(open("$OUT/c5_${arm}_$rep.err").read()[-600:])
PY
  done
done | tee $OUT/c5_gqa_launch_ab.txt
