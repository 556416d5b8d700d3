#!/bin/bash
# round 3, GPU call 7: new LP test, sampling trace (no sort / topk), LP collective latency under rocprofv3, c5 + c2 lines with traffic
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lp.py tests/test_gpu_bench_pins.py -x -q -k "abi_communicator or config_lade or two_gloo" > $OUT/pytest_lp7.log 2>&1; echo "lp tests rc=$?"; tail -4 $OUT/pytest_lp7.log
# sampling step with top-k / top-p: kernel list
rm -rf /tmp/st; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $ROOT/tools/sample_trace.py > /tmp/st.log 2>&1); tail -2 /tmp/st.log
python tools/kstats.py $(find /tmp/st -name "*kernel_stats.csv" | head -1) 60 > $OUT/sample_trace_kernel_stats.txt; grep -i "warp_rows\|sort\|topk\|cumsum\|softmax\|scatter" $OUT/sample_trace_kernel_stats.txt
# the step's collective with ONE rank, inside the LP step (hipGraph segments around it), both forms
for coll in torch abi; do
  rm -rf /tmp/lp_$coll
  (cd /tmp && LADE_LP_COLLECTIVE=$coll timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp_$coll -- python $ROOT/bench.py --gpus 1 --force-lp --layers 8 --steps 32 --warmup 8 --blocks 1 --no-cpu-baseline --no-extras > /tmp/lp_$coll.log 2>&1)
  grep "^{" /tmp/lp_$coll.log | cut -c1-300
  python tools/trace_medians.py $(find /tmp/lp_$coll -name "*kernel_trace.csv" | head -1) > $OUT/lp_${coll}_kernel_medians.txt
  grep -i "nccl\|rccl\|AllGather\|lp_pack\|lp_reduce\|Memcpy\|copy" $OUT/lp_${coll}_kernel_medians.txt | head
done
timeout 1500 python bench.py --config c5 --steps 32 --warmup 8 2> $OUT/bench_c5b.err | grep "^{" > $OUT/bench_c5b.json; echo "c5 rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 2> $OUT/bench_c2b.err | grep "^{" > $OUT/bench_c2b.json; echo "c2 rc=$?"
python - <<P
import json
for f in ("bench_c5b","bench_c2b"):
    d=json.load(open("$OUT/"+f+".json")); r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "roofline", r["frac"], r["launch_us"], r["launch_us_source"], "traffic", r["traffic"], "in_step", r["launch_us_in_step"])
P
