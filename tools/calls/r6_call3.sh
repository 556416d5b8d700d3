#!/bin/bash
# round 6, call 3: the 160-row GEMM class and the ping-pong K loop against the round-5 kernel, isolated (tools/gemm_r6_probe.py); zero padding rows against
# the power cap (tools/clock_probe.py CASES, LADE_DEBUG=gemm_dbg=256); step(T rows) curve with / without the 160-row class (tools/rows_curve.py); the GPU suite
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6c
mkdir -p $OUT
{
MODEL=13b M=150 timeout 900 python tools/gemm_r6_probe.py 2>&1 | grep -v amdgpu.ids
MODEL=13b M=126 timeout 900 python tools/gemm_r6_probe.py 2>&1 | grep -v amdgpu.ids
MODEL=7b M=150 timeout 900 python tools/gemm_r6_probe.py 2>&1 | grep -v amdgpu.ids
MODEL=7b M=120 timeout 900 python tools/gemm_r6_probe.py 2>&1 | grep -v amdgpu.ids
MODEL=7b M=92 timeout 900 python tools/gemm_r6_probe.py 2>&1 | grep -v amdgpu.ids
} | tee $OUT/gemm_r6_probe.txt
{
CASES=76,120 LADE_DEBUG=gemm_dbg=0 timeout 300 python tools/clock_probe.py 2>&1 | grep -v amdgpu.ids
CASES=76,120 LADE_DEBUG=gemm_dbg=256 timeout 300 python tools/clock_probe.py 2>&1 | grep -v amdgpu.ids
} | tee $OUT/zero_pad_probe.txt
{
timeout 900 python tools/rows_curve.py 7b 2>&1 | grep -v amdgpu.ids
LADE_DEBUG=row_classes=r5 timeout 900 python tools/rows_curve.py 7b 1 128 132 144 156 160 180 192 2>&1 | grep -v amdgpu.ids
} | tee $OUT/rows_curve_7b.txt
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
