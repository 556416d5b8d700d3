#!/bin/bash
# round 3, GPU call 2: in-launch split merge - parity + stress tests, whole GPU suite, A/B against the two-launch merge
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bench_pins.py -x -q -s > $OUT/pytest_pins.log 2>&1; echo "pins rc=$?"; grep -v "^$" $OUT/pytest_pins.log | tail -25
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_bench_pins.py > $OUT/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -8 $OUT/pytest_gpu.log
for rep in 1 2; do
  for mg in launch xcd; do
    timeout 200 python tools/attn_bench.py --merge $mg --T 60 120 --P 2016 4096 --splits 0 6 8 2>&1 | grep "T="
    timeout 200 python tools/attn_bench.py --merge $mg --T 60 --P 2016 --H 64 --Hkv 8 --splits 0 4 8 2>&1 | grep "T="
    timeout 200 python tools/attn_bench.py --merge $mg --T 60 --P 128 512 1024 --splits 0 2>&1 | grep "T="
  done
done > $OUT/attn_ab2.txt 2>&1
cat $OUT/attn_ab2.txt
for mg in launch xcd; do
  LADE_ATTN_MERGE=$mg timeout 600 python bench.py --steps 32 --warmup 8 --no-cpu-baseline 2> $OUT/bench_$mg.err | grep "^{" > $OUT/bench_$mg.json
  python - <<P
import json
d=json.load(open("$OUT/bench_$mg.json"))
r=d["roofline"]
print("$mg", d["value"], d["ms_per_step"], "plain", d["plain_decode"]["ms_per_token"], "hot", d["hot_regime"]["value"], "pair in step", r["launch_us_in_step"] and r["launch_us_in_step"]["us"], "iso", r["launch_us_isolated"], "frac", r["frac"])
P
done
