#!/bin/bash
# round 4, call 6: sealed record + host polling (tests, A/B), and the Infinity-Cache read-ahead experiment (csrc/prefetch.hip) at
# 32 / 64 / 128 work-groups, default and non-temporal loads; all alternating on one box
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_hf.py -x -q 2>&1 | tail -5
B="python bench.py --steps 32 --warmup 8 --no-cpu-baseline --blocks 3"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 400 $B 2> $OUT/pf_$name.err | grep "^{" > $OUT/pf_$name.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/pf_$name.json"))
    print("$name", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "plain", d["plain_decode"]["ms_per_token"], "hot", d["hot_regime"]["value"], "prefill", d["prefill"]["tokens_per_s"])
except Exception as e:
    print("$name FAILED", e); print(open("$OUT/pf_$name.err").read()[-1500:])
PY
}
for rep in 1 2; do
  run poll0_$rep LADE_POLL=0
  run poll1_$rep LADE_POLL=1
  run pf64_$rep LADE_PREFETCH=64
  run pf128_$rep LADE_PREFETCH=128
  run pf64nt_$rep LADE_PREFETCH=64 LADE_PREFETCH_NT=1
  run pf32_$rep LADE_PREFETCH=32
done
