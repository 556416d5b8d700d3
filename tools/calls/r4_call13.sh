#!/bin/bash
# round 4, call 13: attention prologue reorder (Q tile first) against the round-3 kernel: isolated pair and in-step, alternating; bench pins
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bench_pins.py tests/test_gpu_kernels.py -x -q -k "attention or attn" 2>&1 | tail -2
for rep in 1 2; do
  echo "== new $rep"; timeout 200 python tools/attn_bench.py --T 60 120 --P 2048 --splits 0 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== old $rep"; (cd _ab_old && timeout 200 python tools/attn_bench.py --T 60 120 --P 2048 --splits 0 2>&1 | grep -v amdgpu.ids | tail -3)
done
for rep in 1 2; do
  echo "== new in-step $rep"; timeout 300 python tools/attn_in_step.py --T 60 --splits 0 2>&1 | grep -v amdgpu.ids | tail -2
  echo "== old in-step $rep"; (cd _ab_old && timeout 300 python tools/attn_in_step.py --T 60 --splits 0 2>&1 | grep -v amdgpu.ids | tail -2)
done
