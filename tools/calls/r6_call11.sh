#!/bin/bash
# round 6, call 11: the ping-pong K loop (EXPERIMENTAL build) where it was never measured - steps / chunks wider than 256 rows, compute bound - against hipBLASLt and the
# lock-step 256-row shapes (tools/gemm_wide_probe.py); + the experimental build's tests once more (16-row tiles included)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6k
export LADE_HIP_LIB=$PWD/lookaheaddecoding_amd/liblade_hip_exp.so
timeout 900 python -m pytest tests/test_gpu_ktile.py -m gpu -q 2>&1 | tail -3 | tee gpurun_out/r6k/pytest_experimental_build.txt
timeout 1500 python tools/gemm_wide_probe.py 7b 420 847 2304 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6k/gemm_wide_pingpong_probe.txt
