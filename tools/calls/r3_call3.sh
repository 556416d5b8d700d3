#!/bin/bash
# round 3, GPU call 3: ticket-based in-launch merge - tests, in-step cost of both merge forms, isolated A/B, timeline
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_bench_pins.py -x -q -k "attention or merge" > $OUT/pytest_pins3.log 2>&1; echo "pins rc=$?"; tail -4 $OUT/pytest_pins3.log
timeout 1200 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_bench_pins.py > $OUT/pytest_gpu3.log 2>&1; echo "gpu suite rc=$?"; tail -4 $OUT/pytest_gpu3.log
timeout 600 python tools/attn_in_step.py --T 60 120 --splits 0 8 > $OUT/attn_in_step3.txt 2>&1; cat $OUT/attn_in_step3.txt
for rep in 1 2; do
  for mg in launch xcd; do
    timeout 200 python tools/attn_bench.py --merge $mg --T 60 120 --P 2016 --splits 6 8 2>&1 | grep "T="
    timeout 200 python tools/attn_bench.py --merge $mg --T 60 --P 2016 --H 64 --Hkv 8 --splits 6 8 2>&1 | grep "T="
  done
done > $OUT/attn_ab3.txt 2>&1
cat $OUT/attn_ab3.txt
L=$ROOT/lookaheaddecoding_amd
for mg in launch xcd; do
LADE_ATTN_DBG=16 LADE_HIP_LIB=$L/liblade_hip_tl.so timeout 120 python tools/attn_bench.py --merge $mg --T 60 --P 2016 --splits 6 --reps 50 2>&1 | grep -v amdgpu.ids
done > $OUT/attn_timeline3.txt 2>&1
cat $OUT/attn_timeline3.txt
