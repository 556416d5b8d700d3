#!/bin/bash
# round 6, call 14: the power model of DESIGN 4.6 checked on the library's compute-bound GEMM (a regime it was not fitted on)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6n
timeout 600 python tools/power_model_check.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6n/power_model_check.txt
