#!/bin/bash
# kernel trace of the 70B-shape bench line (weights K-tile-major only)
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/kt5
(cd /tmp && timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -- python $ROOT/bench.py --config c5 --steps 32 --warmup 8 --blocks 1 --no-cpu-baseline --no-extras > /tmp/kt5.log 2>&1); echo "rc=$?"
grep "^{" /tmp/kt5.log > $OUT/bench_c5_under_rocprof.json
python tools/trace_medians.py $(find /tmp/kt5 -name "*kernel_trace.csv" | head -1) > $OUT/bench_c5_kernel_medians.txt
head -14 $OUT/bench_c5_kernel_medians.txt | cut -c1-150
python -c "
import json; d=json.load(open('$OUT/bench_c5_under_rocprof.json')); print(d['value'], d['ms_per_step'])"
