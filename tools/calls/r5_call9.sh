#!/bin/bash
# round 5, call 9: the lookahead-parallel expectation refreshed on the final build - the reference's default W=60 N=8 G=60 on one rank (bench lp7b) and every rank's
# shard at R = 1 / 2 / 4 / 8 timed on one GPU (tools/lp_curve.py): what `lp_default.expected_speedup_vs_one_rank` in the N > 1 bench line rests on
set -u
cd "$(dirname "$0")/../.."
OUT=$PWD/gpurun_out/r5p
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py --config lp7b --gpus 1 --steps 16 --warmup 4 --no-cpu-baseline --blocks 2 2> $OUT/bench_lp7b.err | grep "^{" > $OUT/bench_lp7b.json
echo "bench lp7b rc=$? $(cut -c1-200 $OUT/bench_lp7b.json)"
timeout 900 python tools/lp_curve.py 7b 60 8 60 2>&1 | grep -v amdgpu.ids > $OUT/lp_curve_7b.txt; tail -12 $OUT/lp_curve_7b.txt | cut -c1-220
