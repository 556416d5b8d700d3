#!/bin/bash
# round 4, call 11: dynamic-NTK RoPE on the HIP step against the reference's own runs
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_hf.py -x -q 2>&1 | tail -12
