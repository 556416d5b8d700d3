#!/bin/bash
# round 5, call 1: first contact of (a) the flash_attn_func adapter + lade_kv_pack_bshd, (b) RoPE + KV append fused into the attention
# launch (bit-identity against the two-launch form), (c) wg_rows as a launch parameter, (d) the in-step attention tuner; then the engine-level
# suites with the fused launch on by default, then c2 A/B: fused off / on (tuner off), then the tuner's own ranking.
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_fused_rope.py -x -q 2>&1 | tail -25
timeout 600 python -m pytest tests/test_gpu_flash_boundary.py -x -q 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "attention or rope or mask" 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_hf.py tests/test_gpu_parity_shapes.py -x -q 2>&1 | tail -8
for rep in 1 2; do
  for v in 0 1; do
    LADE_FUSE_ROPE=$v LADE_ATTN_TUNE=0 timeout 600 python bench.py --config c2 --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/fr_c2_${v}_$rep.err | grep "^{" > $OUT/fr_c2_${v}_$rep.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/fr_c2_${v}_$rep.json"))
    print("c2 fuse_rope=$v rep $rep", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"])
except Exception as e:
    print("c2 fuse=$v $rep FAILED", e); print(open("$OUT/fr_c2_${v}_$rep.err").read()[-2500:])
PY
  done
done
LADE_TUNE_VERBOSE=1 timeout 900 python bench.py --config c2 --steps 32 --warmup 8 --no-cpu-baseline --blocks 3 2> $OUT/tune_c2.err | grep "^{" > $OUT/tune_c2.json
grep "tune-step\] attn" $OUT/tune_c2.err | head -20
python - <<PY
import json
try:
    d=json.load(open("$OUT/tune_c2.json"))
    print("c2 tuned", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "plain", d["plain_decode"]["ms_per_token"], "pair", d["roofline"]["launch_us"], d["roofline"]["frac"])
    print(json.dumps(d["projections"]["in_step_tuning"].get("attn")))
except Exception as e:
    print("c2 tuned FAILED", e); print(open("$OUT/tune_c2.err").read()[-2500:])
PY
