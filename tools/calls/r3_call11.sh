#!/bin/bash
# K-tile-major copies in the engine: new tests, then the c2 bench line with and without them
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ktile.py -x -q 2>&1 | tail -5
for kt in 1 0; do
  LADE_W_KTILE=$kt LADE_TUNE_VERBOSE=1 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3/bench_kt$kt.out 2> gpurun_out/r3/bench_kt$kt.err; echo "bench kt=$kt rc=$?"
  grep "^{" gpurun_out/r3/bench_kt$kt.out > gpurun_out/r3/bench_kt$kt.json
  python - <<P
import json
d=json.load(open("gpurun_out/r3/bench_kt$kt.json")); r=d["roofline"]
print("kt=$kt", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "plain", d["plain_decode"]["ms_per_token"], "hot", d["hot_regime"]["value"], d["hot_regime"]["step_compression"], "forced", d["hot_regime_forced"]["value"], "prefill", d["prefill"]["tokens_per_s"])
print("   roofline", r["frac"], r["launch_us"], r["launch_us_source"], "| layout:", d["config"]["weight_layout"][:60])
P
  grep "^\[tune\]" gpurun_out/r3/bench_kt$kt.err | cut -c1-200 | head -12
done
