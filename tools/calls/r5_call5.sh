#!/bin/bash
# round 5, call 4: the producer mode of the fused RoPE (dedicated work-groups do the RoPE + append once per KV head, in-launch hand-off through
# write-through stores + a per-head flag): bit-identity tests, then c2 A/B in alternation (two launches / producer mode, tuner off), then the tuner's ranking
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fused_rope.py -x -q 2>&1 | tail -25
for rep in 1 2; do
  for v in 0 2; do
    LADE_FUSE_ROPE=$v LADE_ATTN_TUNE=0 timeout 600 python bench.py --config c2 --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/pm2_c2_${v}_$rep.err | grep "^{" > $OUT/pm2_c2_${v}_$rep.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/pm2_c2_${v}_$rep.json"))
    print("c2 fuse_rope=$v rep $rep", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"])
except Exception as e:
    print("c2 fuse=$v $rep FAILED", e); print(open("$OUT/pm2_c2_${v}_$rep.err").read()[-2500:])
PY
  done
done
LADE_TUNE_VERBOSE=1 timeout 900 python bench.py --config c2 --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 2 2> $OUT/pm2_tune_c2.err | grep "^{" > $OUT/pm2_tune_c2.json
grep "tune-step\] attn" $OUT/pm2_tune_c2.err | cut -c1-500
