#!/bin/bash
# round 3, GPU call 4: slimmed attention prologue / tail - suite, in-step cost, bench line with the new fields
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu4.log 2>&1; echo "gpu suite rc=$?"; tail -4 $OUT/pytest_gpu4.log
timeout 600 python tools/attn_in_step.py --T 60 120 --splits 0 > $OUT/attn_in_step4.txt 2>&1; cat $OUT/attn_in_step4.txt
for rep in 1 2; do
    timeout 200 python tools/attn_bench.py --T 60 120 --P 128 2016 4096 --splits 0 2>&1 | grep "T="
    timeout 200 python tools/attn_bench.py --T 60 --P 2016 --H 64 --Hkv 8 --splits 0 8 2>&1 | grep "T="
done > $OUT/attn_ab4.txt 2>&1
cat $OUT/attn_ab4.txt
L=$ROOT/lookaheaddecoding_amd
LADE_ATTN_DBG=16 LADE_HIP_LIB=$L/liblade_hip_tl.so timeout 120 python tools/attn_bench.py --T 60 --P 2016 --splits 6 --reps 50 2>&1 | grep -v amdgpu.ids > $OUT/attn_timeline4.txt
cat $OUT/attn_timeline4.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2> $OUT/bench4.err | grep "^{" > $OUT/bench4.json; echo "bench rc=$?"; tail -3 $OUT/bench4.err
python - <<P
import json
d=json.load(open("$OUT/bench4.json"))
r=d["roofline"]
print(d["value"], d["ms_per_step"], d["spread"])
print("plain", d["plain_decode"])
print("hot live", d["hot_regime"])
print("hot forced", d["hot_regime_forced"]["value"], d["hot_regime_forced"]["ms_per_step"])
print("roofline", r["frac"], r["launch_us"], r["launch_us_source"], r["launch_us_in_step"], r["launch_us_graph_delta"], r["launch_us_isolated"], r["traffic"])
print(d["config"])
P
