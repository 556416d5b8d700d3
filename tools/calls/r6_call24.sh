#!/bin/bash
# round 6, call 24: the GPU suite's summary line on the final library
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6x
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tee gpurun_out/r6x/pytest_gpu.txt
