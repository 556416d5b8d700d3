#!/bin/bash
# round 5, call 16: what bounds the 128-row class - the ablation bits one by one at 13B / 120 rows (1 = no stores, 4 = no LDS reads / MFMA,
# 128 = no activation traffic), and all-zero operands (same instruction stream, less switching power: is the class clock-bound?)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5q
{
for rep in 1 2; do
for d in 0 1 4 5 128 129 132 133; do
  LADE_GEMM_DBG=$d MODEL=13b M=120 python tools/gemm_ingest_probe.py 2>&1 | tail -1
done
ZERO=1 MODEL=13b M=120 python tools/gemm_ingest_probe.py 2>&1 | tail -1 | sed 's/^/ZERO operands: /'
done
} | tee gpurun_out/r5q/gemm_ablation_13b_120.txt
