#!/bin/bash
# Produces the round's evidence files under gpurun_out/r4p/ (run on the GPU box from the repo root: bash tools/make_profiles_r4.sh);
# the summaries are then copied into profiles/ as r4_* (see profiles/README.md).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4p
mkdir -p $OUT
export TMPDIR=/tmp
# 1. bench lines, un-profiled: the default configuration with the CPU baseline, then the other configurations (bf16), then f16
for c in c2 c3 c4 c5; do
    LADE_DEBUG=tune_verbose timeout 1500 python bench.py --config $c --steps 32 --warmup 8 2> $OUT/bench_$c.err | grep "^{" > $OUT/bench_$c.json
    echo "bench $c rc=$? $(cut -c1-150 $OUT/bench_$c.json)"
done
for c in c2 c4; do
    timeout 1500 python bench.py --config $c --dtype f16 --steps 32 --warmup 8 --no-cpu-baseline 2> $OUT/bench_${c}_f16.err | grep "^{" > $OUT/bench_${c}_f16.json
    echo "bench $c f16 rc=$? $(cut -c1-150 $OUT/bench_${c}_f16.json)"
done
# 2. kernel trace of the default bench, taken apart launch by launch
rm -rf /tmp/kt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $ROOT/bench.py --steps 32 --warmup 8 --blocks 1 --no-cpu-baseline --no-extras > /tmp/kt.log 2>&1)
grep "^{" /tmp/kt.log > $OUT/bench_c2_under_rocprof.json
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/bench_c2_kernel_stats.csv
python tools/trace_medians.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) --steps > $OUT/bench_c2_kernel_medians.txt
# 3. counters of the attention pair at the bench's own launch shapes (SKIP_PMC=1: the attention kernels did not change since the last pass)
if [ "${SKIP_PMC:-0}" != "1" ]; then
P_END=$(python -c "import json; print(json.load(open('$OUT/bench_c2.json'))['config']['kv_len_end'])")
bash tools/attn_pmc.sh $OUT/attn_pmc.json $P_END > $OUT/attn_pmc.log 2>&1; tail -4 $OUT/attn_pmc.log
fi
# 4. the lookahead-parallel regime: the reference's default W=60 N=8 G=60 on one rank, and every rank's shard at R = 1 / 2 / 4 / 8
if [ "${SKIP_LP:-0}" != "1" ]; then
timeout 900 python bench.py --config lp7b --gpus 1 --steps 16 --warmup 4 --no-cpu-baseline --blocks 2 2> $OUT/bench_lp7b.err | grep "^{" > $OUT/bench_lp7b.json
echo "bench lp7b rc=$? $(cut -c1-150 $OUT/bench_lp7b.json)"
timeout 900 python tools/lp_curve.py 7b 60 8 60 2>&1 | grep -v amdgpu.ids > $OUT/lp_curve_7b.txt; tail -8 $OUT/lp_curve_7b.txt | cut -c1-200
fi
ls -la $OUT
