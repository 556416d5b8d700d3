#!/bin/bash
# round 6, call 21: the GPU suite and smoke() on the final library (cache policies as compile-time constants)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6u
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r6u/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r6u/smoke.txt
