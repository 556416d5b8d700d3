#!/bin/bash
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
timeout 500 python tools/lm_head_probe.py > gpurun_out/r3/lm_head_probe.txt 2>&1; echo "lm_head probe rc=$?"; grep -v "^      " gpurun_out/r3/lm_head_probe.txt
for cons in 0 8 4; do
  if [ $cons = 0 ]; then unset LADE_TUNE_CONSUMER; else export LADE_TUNE_CONSUMER=$cons; fi
  LADE_TUNE_VERBOSE=1 timeout 600 python bench.py --no-cpu-baseline --blocks 3 > gpurun_out/r3/bench_cons$cons.out 2> gpurun_out/r3/bench_cons$cons.err; echo "bench consumer=$cons rc=$?"
  grep "^{" gpurun_out/r3/bench_cons$cons.out > gpurun_out/r3/bench_cons$cons.json
  python - <<P
import json
d=json.load(open("gpurun_out/r3/bench_cons$cons.json"))
print("consumer=$cons", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "plain", d["plain_decode"]["ms_per_token"], "hot", d["hot_regime"]["value"])
P
  grep "^\[tune\].*:\(64\|32\) " gpurun_out/r3/bench_cons$cons.err | cut -c1-110
done
