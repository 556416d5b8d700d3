#!/bin/bash
# round 6, call 10: power / clock while WHOLE steps replay (tools/step_power_probe.py: is the step itself power limited above 96 rows?), kernel trace of the c4 line
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6j
mkdir -p $OUT
export TMPDIR=/tmp
{
timeout 900 python tools/step_power_probe.py 13b 1 60 92 120 150 180 2>&1 | grep -v amdgpu.ids
timeout 900 python tools/step_power_probe.py 7b 1 60 92 120 150 180 2>&1 | grep -v amdgpu.ids
} | tee $OUT/step_power_probe.txt
rm -rf /tmp/kt4
ROOT=$PWD
(cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt4 -- python $ROOT/bench.py --config c4 --steps 32 --warmup 8 --blocks 1 --no-cpu-baseline --no-extras > /tmp/kt4.log 2>&1)
python tools/trace_medians.py $(find /tmp/kt4 -name "*kernel_trace.csv" | head -1) --steps > $OUT/bench_c4_kernel_medians.txt
grep -A14 "per kernel name inside" $OUT/bench_c4_kernel_medians.txt | cut -c1-120
