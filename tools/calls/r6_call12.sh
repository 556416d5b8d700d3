#!/bin/bash
# round 6, call 12: the GPU suite of the final tree (new: the generate-surface speed test, the 16-row-tile refusal test)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6l
timeout 900 python -m pytest tests/test_gpu_hf.py -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r6l/pytest_hf.txt
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r6l/pytest_gpu.txt
