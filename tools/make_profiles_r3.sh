#!/bin/bash
# Produces the round's evidence files under gpurun_out/r3/ (run on the GPU box from the repo root: bash tools/make_profiles_r3.sh);
# the summaries are then copied into profiles/ (see profiles/README.md).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
# 0. the GPU suite first: nothing below is worth keeping from a build that fails it
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "gpu suite rc=$rc"; tail -5 $OUT/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
# 1. bench lines, un-profiled: the default configuration with the CPU baseline, then the other configurations
for c in c2 c3 c4 c5; do
    timeout 1200 python bench.py --config $c --steps 32 --warmup 8 2> $OUT/bench_$c.err | grep "^{" > $OUT/bench_$c.json
    echo "bench $c rc=$? $(cut -c1-160 $OUT/bench_$c.json)"
done
# 2. kernel trace + stats of the default bench (profiled run: its own bench line is kept beside the trace)
rm -rf /tmp/kt
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $ROOT/bench.py --steps 32 --warmup 8 --blocks 1 --no-cpu-baseline --no-extras > /tmp/kt.log 2>&1)
grep "^{" /tmp/kt.log > $OUT/bench_c2_under_rocprof.json
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/bench_c2_kernel_stats.csv
python tools/trace_medians.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) > $OUT/bench_c2_kernel_medians.txt
python tools/kstats.py $OUT/bench_c2_kernel_stats.csv 30 > $OUT/bench_c2_kernel_stats.txt
head -30 $OUT/bench_c2_kernel_medians.txt
# 3. counters of the attention pair at the bench's own launch shapes
P_END=$(python -c "import json; print(json.load(open('$OUT/bench_c2.json'))['config']['kv_len_end'])")
bash tools/attn_pmc.sh $OUT/attn_pmc.json $P_END > $OUT/attn_pmc.log 2>&1; tail -4 $OUT/attn_pmc.log
# 4. microbenchmarks
python tools/attn_bench.py --T 60 120 --P 128 512 1024 2048 4096 --splits 0 > $OUT/attn_sweep.txt 2>&1
python tools/attn_in_step.py --T 60 120 --splits 0 > $OUT/attn_in_step.txt 2>&1
python tools/attn_in_step.py --model codellama-13b --layers 6 --T 120 --splits 0 >> $OUT/attn_in_step.txt 2>&1
cat $OUT/attn_in_step.txt
# 5. weight layout probes (row-major against K-tile-major): only with PROBES=1 (another ~1 min)
if [ "${PROBES:-0}" = 1 ]; then
    python tools/gemm_ktile_probe.py 7b 60 30 120 > $OUT/gemm_ktile_probe.txt 2>&1
    python tools/gemm_ktile_probe.py 13b 120 >> $OUT/gemm_ktile_probe.txt 2>&1
    python tools/gemm_ktile_probe.py 70b 60 30 > $OUT/gemm_ktile_probe_70b.txt 2>&1
    python tools/lm_head_probe.py > $OUT/lm_head_probe.txt 2>&1
    grep -h "layer sum\|rows=" $OUT/gemm_ktile_probe.txt $OUT/gemm_ktile_probe_70b.txt $OUT/lm_head_probe.txt
fi
ls -la $OUT | head -60
