#!/bin/bash
# round 5, call 13: kernel trace of the c4 bench line on the build with the wide add + norm blocks (compare profiles/r5_bench_c4_kernel_medians.txt)
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5q
export TMPDIR=/tmp
mkdir -p $OUT
for c in c4; do
  rm -rf /tmp/kt_$c
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$c -- python $ROOT/bench.py --config $c --steps 32 --warmup 8 --blocks 1 --no-cpu-baseline --no-extras > /tmp/kt_$c.log 2>&1)
  grep "^{" /tmp/kt_$c.log | cut -c1-160
  python tools/trace_medians.py $(find /tmp/kt_$c -name "*kernel_trace.csv" | head -1) --steps > $OUT/bench_${c}_kernel_medians_wide_norm.txt
  awk '/^steady step: [0-9]+ of/{f=1} f' $OUT/bench_${c}_kernel_medians_wide_norm.txt | sed -n '1,24p' | cut -c1-150
  grep -A12 "per kernel name inside" $OUT/bench_${c}_kernel_medians_wide_norm.txt | cut -c1-120
done
