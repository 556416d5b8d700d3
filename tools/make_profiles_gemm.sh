#!/bin/bash
# Refresh of the evidence that depends on the GEMM kernel only (the attention kernels are untouched: their counters and sweeps stay):
# bench lines of the four configurations, kernel trace + stats of the default bench, GEMM counters, the nt on / off A/B.
# Run on the GPU box from the repo root: bash tools/make_profiles_gemm.sh
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r2
mkdir -p $OUT $ROOT/gpurun_out/pmc
export TMPDIR=/tmp
for c in c2 c3 c4 c5; do
    timeout 900 python bench.py --config $c --steps 32 --warmup 8 2> $OUT/bench_$c.err | grep "^{" > $OUT/bench_$c.json
    echo "bench $c rc=$? $(cut -c1-140 $OUT/bench_$c.json)"
done
rm -rf /tmp/kt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python $ROOT/bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extras > /tmp/kt.log 2>&1)
grep "^{" /tmp/kt.log > $OUT/bench_c2_under_rocprof.json
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $OUT/bench_c2_kernel_stats.csv
python tools/trace_medians.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) > $OUT/bench_c2_kernel_medians.txt
python tools/kstats.py $OUT/bench_c2_kernel_stats.csv 30 > $OUT/bench_c2_kernel_stats.txt
run() {   # label, counters, command...
    local label=$1 ctr=$2; shift 2
    rm -rf /tmp/prof_$label
    (cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/prof_$label -- "$@" > $ROOT/gpurun_out/pmc/$label.log 2>&1)
    local f=$(find /tmp/prof_$label -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $ROOT/gpurun_out/pmc/$label.csv || echo "no counter csv for $label"
}
G128="env M=128 python $ROOT/tools/gemm_flags.py"
G60="env M=60 python $ROOT/tools/gemm_flags.py"
MF="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
LD="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
mkdir -p $OUT/pmc
for w in G128 G60; do
    cmd=${!w}
    run ${w}_mfma "$MF" $cmd
    run ${w}_lds "$LD" $cmd
    run ${w}_fetch "FETCH_SIZE" $cmd
    run ${w}_write "WRITE_SIZE" $cmd
    python tools/pmc_table.py $OUT/pmc/$w.json mfma=gpurun_out/pmc/${w}_mfma.csv lds=gpurun_out/pmc/${w}_lds.csv fetch=gpurun_out/pmc/${w}_fetch.csv write=gpurun_out/pmc/${w}_write.csv --trim $OUT/pmc/csv
done
rm -rf gpurun_out/pmc
for d in 0 16; do LADE_DEBUG=gemm_dbg=$d python tools/gemm_flags.py 2>&1 | tail -1; LADE_DEBUG=gemm_dbg=$d M=128 python tools/gemm_flags.py 2>&1 | tail -1; done > $OUT/gemm_nt_ab.txt
for d in 0 1 4 5; do LADE_DEBUG=gemm_dbg=$d python tools/gemm_ingest_probe.py 2>&1 | tail -1; done > $OUT/gemm_ingest.txt
ls -la $OUT | head -40
