#!/bin/bash
# round 4, call 2: the ring-depth launch parameter + wide rows + weight memory plan (GPU tests), then c2 with the ring-aware autotune
# (verbose: what it picked) and c5 (prefill with the row-major originals kept while the HBM budget lasts)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ktile.py tests/test_gpu_kernels.py -x -q 2>&1 | tail -15
LADE_TUNE_VERBOSE=1 timeout 400 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --blocks 3 2> $OUT/c2_ring.err | grep "^{" > $OUT/c2_ring.json
grep "^\[tune\]" $OUT/c2_ring.err | cut -c1-330
python - <<PY
import json
d=json.load(open("$OUT/c2_ring.json"))
print("c2", d["value"], d["ms_per_step"], d.get("spread",{}), d.get("plain_decode"), d.get("roofline",{}).get("launch_us"))
PY
timeout 900 python bench.py --config c5 --steps 16 --warmup 4 --no-cpu-baseline --no-extras --blocks 2 2> $OUT/c5_plan.err | grep "^{" > $OUT/c5_plan.json
python - <<PY
import json
d=json.load(open("$OUT/c5_plan.json"))
print("c5", d["value"], d["ms_per_step"], d.get("prefill"), d["config"].get("weight_layout"))
PY
tail -5 $OUT/c5_plan.err
