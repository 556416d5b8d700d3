#!/bin/bash
# round 4, call 12: rope kernel after sizing its partial rounds (glue bench), GPU tests of kv kernels, c2 bench with the back-to-back measurement
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -2
timeout 200 python tools/glue_bench.py --T 60 2>&1 | grep -v amdgpu.ids | tee $OUT/glue_bench.txt | tail -14
timeout 600 python bench.py --steps 32 --warmup 8 --no-cpu-baseline --blocks 3 2> $OUT/b12.err | grep "^{" > $OUT/b12.json
python - <<PY
import json
d=json.load(open("$OUT/b12.json"))
print("c2", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "gpu_only", d["step_gpu_only"], "plain", d["plain_decode"]["ms_per_token"])
PY
