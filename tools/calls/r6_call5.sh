#!/bin/bash
# round 6, call 5: the GPU suite on the tree with the 160-row class, LADE_DEBUG, the tune file; then the decision table shipped with the package
# (tools/make_tune_table.py: 7B bf16 / f16, 13B bf16 - every row class tuned once, here)
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6e
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
LADE_DEBUG=tune_verbose timeout 3000 python tools/make_tune_table.py $PWD/$OUT/tuned.json 7b:bf16 13b:bf16 7b:f16 2> $OUT/make_tune_table.err | tee $OUT/make_tune_table.txt
tail -3 $OUT/make_tune_table.err
