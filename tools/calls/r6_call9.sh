#!/bin/bash
# round 6, call 9: the 16-row-granular GEMM (csrc/gemm16.hpp, probe state: split-K partials) against the padded 32-row classes, isolated
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6i
{
MODEL=7b M=76 timeout 900 python tools/gemm16_probe.py 2>&1 | grep -v amdgpu.ids
MODEL=7b M=104 timeout 900 python tools/gemm16_probe.py 2>&1 | grep -v amdgpu.ids
MODEL=7b M=138 timeout 900 python tools/gemm16_probe.py 2>&1 | grep -v amdgpu.ids
MODEL=13b M=138 timeout 900 python tools/gemm16_probe.py 2>&1 | grep -v amdgpu.ids
MODEL=13b M=174 timeout 900 python tools/gemm16_probe.py 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r6i/gemm16_probe.txt
