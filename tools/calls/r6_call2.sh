#!/bin/bash
# round 6, call 2: the GPU suite on the tree of commit 56ea120 (epilogue split out of the ring kernel, RA kernel, EXPERIMENTAL switch, frozen attention launch) and the
# bench lines c2 / c4 it gives - the baseline of this round's GEMM work
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6b
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee $OUT/pytest_gpu.txt
timeout 600 python bench.py 2> $OUT/bench_c2.err | grep "^{" > $OUT/bench_c2.json; tail -c 600 $OUT/bench_c2.json
timeout 900 python bench.py --config c4 --no-cpu-baseline 2> $OUT/bench_c4.err | grep "^{" > $OUT/bench_c4.json; tail -c 600 $OUT/bench_c4.json
