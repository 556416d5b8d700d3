#!/bin/bash
# round 5, call 17: kernel trace of the c3 bench line (sampling, temperature 0.8): what the sampling step adds to the greedy one
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5q
export TMPDIR=/tmp
mkdir -p $OUT
c=c3
rm -rf /tmp/kt_$c
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$c -- python $ROOT/bench.py --config $c --steps 32 --warmup 8 --blocks 1 --no-cpu-baseline --no-extras > /tmp/kt_$c.log 2>&1)
grep "^{" /tmp/kt_$c.log | cut -c1-160
python tools/trace_medians.py $(find /tmp/kt_$c -name "*kernel_trace.csv" | head -1) --steps > $OUT/bench_${c}_kernel_medians.txt
grep -n "steady" $OUT/bench_${c}_kernel_medians.txt | head
grep -v "gemm_skinny\|attn_fwd\|attn_combine\|rope_kv\|rmsnorm_kernel<lade::BF16, true" $OUT/bench_${c}_kernel_medians.txt | head -60 | cut -c1-170
