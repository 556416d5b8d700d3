#!/bin/bash
# Produces the round's evidence files under gpurun_out/r2/ (run on the GPU box from the repo root: bash tools/make_profiles.sh);
# the summaries are then copied into profiles/ by hand (see profiles/README.md).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2
mkdir -p $OUT
export TMPDIR=/tmp
# 1. bench lines, un-profiled: the default configuration with the CPU baseline, the other configurations
for c in c2 c3 c4 c5; do
    timeout 900 python bench.py --config $c --steps 32 --warmup 8 2> $OUT/bench_$c.err | grep "^{" > $OUT/bench_$c.json
    echo "bench $c rc=$? $(cut -c1-140 $OUT/bench_$c.json)"
done
# 2. kernel trace + stats of the default bench (profiled run: its own bench line is kept beside the trace)
rm -rf /tmp/kt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $ROOT/bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extras > /tmp/kt.log 2>&1)
grep "^{" /tmp/kt.log > $OUT/bench_c2_under_rocprof.json
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/bench_c2_kernel_stats.csv
python tools/trace_medians.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) > $OUT/bench_c2_kernel_medians.txt
python tools/kstats.py $OUT/bench_c2_kernel_stats.csv 30 > $OUT/bench_c2_kernel_stats.txt
# 3. counters
bash tools/collect_pmc.sh > $OUT/collect_pmc.log 2>&1
mkdir -p $OUT/pmc
for w in A7 A7B A70 G128 G60; do
    python tools/pmc_table.py $OUT/pmc/$w.json mfma=gpurun_out/pmc/${w}_mfma.csv lds=gpurun_out/pmc/${w}_lds.csv fetch=gpurun_out/pmc/${w}_fetch.csv write=gpurun_out/pmc/${w}_write.csv --trim $OUT/pmc/csv
done
rm -rf gpurun_out/pmc
# 4. microbenchmarks
python tools/attn_bench.py --T 60 120 --P 128 512 1024 2048 4096 --splits 0 > $OUT/attn_sweep.txt 2>&1
for d in 0 16; do LADE_DEBUG=gemm_dbg=$d python tools/gemm_flags.py 2>&1 | tail -1; LADE_DEBUG=gemm_dbg=$d M=128 python tools/gemm_flags.py 2>&1 | tail -1; done > $OUT/gemm_nt_ab.txt
ls -la $OUT | head -40
