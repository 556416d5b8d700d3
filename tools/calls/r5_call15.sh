#!/bin/bash
# round 5, call 15: every built work-group shape x split count x ring depth, isolated, for the four projections of the 13B shape at 120 rows
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5q
for p in qkv o gate_up down; do
  MODEL=13b M=120 PROJ=$p timeout 600 python tools/gemm_shape_sweep.py 2>&1 | tail -17
done | tee gpurun_out/r5q/gemm_shape_sweep_13b_120.txt
