#!/bin/bash
# the four bench lines again (bench.py's step-time difference now comes from alternating block pairs); kernels unchanged since make_profiles_r3.sh
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
for c in c2 c3 c4 c5; do
    timeout 1200 python bench.py --config $c --steps 32 --warmup 8 2> $OUT/bench_$c.err | grep "^{" > $OUT/bench_$c.json
    echo "bench $c rc=$? $(cut -c1-160 $OUT/bench_$c.json)"
done
