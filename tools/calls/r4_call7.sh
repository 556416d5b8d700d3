#!/bin/bash
# round 4, call 7: whole GPU suite (f16 full-width cases are new), kernel trace of the c2 step taken apart launch by launch,
# the default bench line (mid_regime is new), the lookahead-parallel curve projection at the reference's default W=60 N=8 G=60
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
rm -f $ROOT/gpurun_out/envelope_ratios.jsonl
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -6 $OUT/pytest_gpu.log
cp $ROOT/gpurun_out/envelope_ratios.jsonl $OUT/ 2>/dev/null
rm -rf /tmp/kt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $ROOT/bench.py --steps 32 --warmup 8 --blocks 1 --no-cpu-baseline --no-extras > /tmp/kt.log 2>&1); echo "trace rc=$?"
grep "^{" /tmp/kt.log > $OUT/bench_c2_under_rocprof.json
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/bench_c2_kernel_stats.csv
python tools/trace_medians.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) --steps > $OUT/bench_c2_kernel_medians.txt
sed -n '/^steady step/,$p' $OUT/bench_c2_kernel_medians.txt | cut -c1-150 | head -150
timeout 900 python bench.py --steps 32 --warmup 8 2> $OUT/bench_c2.err | grep "^{" > $OUT/bench_c2.json
python - <<PY
import json
d=json.load(open("$OUT/bench_c2.json"))
print("c2", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "plain", d["plain_decode"], "\nmid", d["mid_regime"], "\nhot", d["hot_regime"]["value"], d["hot_regime"]["step_compression"], "roofline", d["roofline"]["launch_us"], d["roofline"]["frac"], "cpu", d["cpu_baseline"])
PY
timeout 600 python tools/lp_curve.py 7b 60 8 60 8 2>&1 | grep -v amdgpu.ids | tee $OUT/lp_curve_7b.txt | cut -c1-330
