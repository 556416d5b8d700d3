#!/bin/bash
# round 4, call 10 (last form): the post-step kernel alone, this build against the round-3 build; integer-kernel and end-to-end tests
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 300 python tools/post_step_bench.py --old _ab_old/lookaheaddecoding_amd/liblade_hip.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/post_step_bench_final.txt
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -2
