#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ktile.py -x -q 2>&1 | tail -5
LADE_TUNE_VERBOSE=1 timeout 900 python bench.py --config c5 --no-cpu-baseline --blocks 3 > gpurun_out/r3/bench_c5_only.out 2> gpurun_out/r3/bench_c5_only.err; echo "bench c5 rc=$?"
grep "^{" gpurun_out/r3/bench_c5_only.out > gpurun_out/r3/bench_c5_only.json
python - <<P
import json
d=json.load(open("gpurun_out/r3/bench_c5_only.json")); r=d["roofline"]
print("c5", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "prefill", d["prefill"]["tokens_per_s"], "stream", d["step_stream"]["frac"])
print("   roofline", r["frac"], r["launch_us"], r["launch_us_source"][:50], "|", d["config"]["weight_layout"][:70])
P
grep "^\[tune\].*:64 " gpurun_out/r3/bench_c5_only.err | cut -c1-130
tail -3 gpurun_out/r3/bench_c5_only.err | cut -c1-300
