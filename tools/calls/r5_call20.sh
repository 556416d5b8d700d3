#!/bin/bash
# round 5, call 20: kernel traces of the round-4 tree and the current tree on ONE box (the current one is 1.2 % slower at config 2 in call 19): where
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5q
export TMPDIR=/tmp
mkdir -p $OUT
for v in r4 r5 r4 r5; do
  d=$ROOT; [ $v = r4 ] && d=$ROOT/_ab_r4
  rm -rf /tmp/kt_$v
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$v -- python $d/bench.py --steps 32 --warmup 8 --blocks 1 --no-cpu-baseline --no-extras > /tmp/kt_$v.log 2>&1)
  grep "^{" /tmp/kt_$v.log | cut -c1-140
  python tools/trace_medians.py $(find /tmp/kt_$v -name "*kernel_trace.csv" | head -1) --steps > $OUT/trace_tree_$v.txt
  grep "steady step: first" $OUT/trace_tree_$v.txt
done
