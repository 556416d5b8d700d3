#!/bin/bash
# round 4, call 9: post-step after the fence removal (trace), c3 / c4 / c5 bench lines with the in-step tuner
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -2
rm -rf /tmp/kt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $ROOT/bench.py --steps 16 --warmup 4 --blocks 1 --no-cpu-baseline --no-extras > /tmp/kt.log 2>&1); echo "trace rc=$?"
python tools/trace_medians.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) --steps > $OUT/trace3.txt
grep -E "greedy_post_step|build_inputs|kv_commit|steady step" $OUT/trace3.txt | cut -c1-150 | tail -7
for c in c3 c4 c5; do
  LADE_TUNE_VERBOSE=1 timeout 1500 python bench.py --config $c --steps 32 --warmup 8 --no-cpu-baseline --blocks 3 2> $OUT/bench_$c.err | grep "^{" > $OUT/bench_$c.json
  python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_$c.json"))
    print("$c", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "T", d["config"]["tokens_per_step_T"], "plain", (d.get("plain_decode") or {}).get("ms_per_token"), "mid", (d.get("mid_regime") or {}).get("step_compression"), (d.get("mid_regime") or {}).get("speedup_vs_plain"), "hot", (d.get("hot_regime") or {}).get("value"), (d.get("hot_regime") or {}).get("step_compression"), "prefill", d["prefill"]["tokens_per_s"], "pair", d["roofline"]["launch_us"], d["roofline"]["frac"], "stream", d["step_stream"]["frac"])
except Exception as e:
    print("$c FAILED", e); print(open("$OUT/bench_$c.err").read()[-1200:])
PY
  grep "tune-step" $OUT/bench_$c.err | cut -c1-170 | head -12
done
