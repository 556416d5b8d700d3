#!/bin/bash
# final check of the round: the whole GPU suite, smoke(), the default bench line
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_final.log 2>&1; echo "gpu suite rc=$?"; tail -4 $OUT/pytest_gpu_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > $OUT/bench_final.out 2> $OUT/bench_final.err; echo "bench rc=$?"; grep "^{" $OUT/bench_final.out > $OUT/bench_final.json
python - <<P
import json
d=json.load(open("$OUT/bench_final.json")); r=d["roofline"]
print(d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "plain", d["plain_decode"]["ms_per_token"], "hot", d["hot_regime"]["value"], d["hot_regime"]["step_compression"], "forced", d["hot_regime_forced"]["value"])
print("roofline", r["frac"], r["launch_us"], r["launch_us_source"], "traffic", r["traffic"], r["algorithmic_bytes"], "in_step", r["launch_us_in_step"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
P
