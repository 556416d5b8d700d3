#!/bin/bash
# round 4, call 18: kernel trace of the step with the fused tail (embedding lookup in the first norm, argmax in the lm_head GEMM)
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $ROOT/bench.py --steps 32 --warmup 8 --blocks 1 --no-cpu-baseline --no-extras > /tmp/kt.log 2>&1)
T=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python tools/trace_medians.py $T --steps > $OUT/ft_kernel_medians.txt 2>&1
grep -A16 "per kernel name inside" $OUT/ft_kernel_medians.txt
awk '/^steady step: [0-9]+ of/{f=1} f' $OUT/ft_kernel_medians.txt | head -8
awk '/^steady step: [0-9]+ of/{f=1} f' $OUT/ft_kernel_medians.txt | grep -B8 "greedy_post_step" | tail -10
grep "steady step: first" $OUT/ft_kernel_medians.txt
