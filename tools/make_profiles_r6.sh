#!/bin/bash
# Produces the round's evidence files under gpurun_out/r6p/ (run on the GPU box from the repo root: bash tools/make_profiles_r6.sh);
# the summaries are then copied into profiles/ as r6_* (see profiles/README.md).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r6p
mkdir -p $OUT
export TMPDIR=/tmp
# 1. bench lines, un-profiled: the default configuration with the CPU baseline, then the other configurations (bf16), then f16
for c in c2 c4 c5 c3; do
    extra=""; [ "$c" != "c2" ] && [ "${CPU_ALL:-0}" != "1" ] && extra="--no-cpu-baseline"
    LADE_DEBUG=tune_verbose timeout 1500 python bench.py --config $c --steps 32 --warmup 8 $extra 2> $OUT/bench_$c.err | grep "^{" > $OUT/bench_$c.json
    echo "bench $c rc=$? $(cut -c1-150 $OUT/bench_$c.json)"
    grep "tune-step\] attn" $OUT/bench_$c.err | cut -c1-400
done
if [ "${SKIP_F16:-0}" != "1" ]; then
for c in c2; do
    timeout 1500 python bench.py --config $c --dtype f16 --steps 32 --warmup 8 --no-cpu-baseline 2> $OUT/bench_${c}_f16.err | grep "^{" > $OUT/bench_${c}_f16.json
    echo "bench $c f16 rc=$? $(cut -c1-150 $OUT/bench_${c}_f16.json)"
done
fi
# 2. the driver's own invocation (python3 bench.py --gpus 1 --steps 20 --warmup 5), as the round-end run will make it
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench_driver_form.err | grep "^{" > $OUT/bench_driver_form.json
echo "bench driver form rc=$? $(cut -c1-150 $OUT/bench_driver_form.json)"
# 3. kernel trace of the default bench, taken apart launch by launch
rm -rf /tmp/kt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $ROOT/bench.py --steps 32 --warmup 8 --blocks 1 --no-cpu-baseline --no-extras > /tmp/kt.log 2>&1)
grep "^{" /tmp/kt.log > $OUT/bench_c2_under_rocprof.json
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/bench_c2_kernel_stats.csv
python tools/trace_medians.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) --steps > $OUT/bench_c2_kernel_medians.txt
# 4. counters of the attention pair at the bench's own launch shapes and the launch parameters the in-step tuner chose in the lines above
if [ "${SKIP_PMC:-0}" != "1" ]; then
P_END=$(python -c "import json; print(json.load(open('$OUT/bench_c2.json'))['config']['kv_len_end'])")
RAW_DIR=$OUT/pmc bash tools/attn_pmc.sh $OUT/attn_pmc.json $P_END $OUT/bench_c2.json $OUT/bench_c4.json $OUT/bench_c5.json > $OUT/attn_pmc.log 2>&1; tail -6 $OUT/attn_pmc.log
fi
ls -la $OUT
