#!/bin/bash
# round 4, call 8: activation layout probe (row-major vs K-tile-major A), alternating; the post-step kernel after the load batching
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
  for v in 0 1; do
    LADE_GEMM_AKT=$v timeout 300 python tools/gemm_akt_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/gemm_akt_probe.txt | grep "layer sum"
  done
done
grep "AKT=1 M=60" $OUT/gemm_akt_probe.txt | head -8; grep "AKT=0 M=60" $OUT/gemm_akt_probe.txt | head -8
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -3
rm -rf /tmp/kt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python $ROOT/bench.py --steps 16 --warmup 4 --blocks 1 --no-cpu-baseline --no-extras > /tmp/kt.log 2>&1); echo "trace rc=$?"
python tools/trace_medians.py $(find /tmp/kt -name "*kernel_trace.csv" | head -1) --steps > $OUT/trace2.txt
grep -E "greedy_post_step|build_inputs|kv_commit|argmax|gather_rows|steady step" $OUT/trace2.txt | cut -c1-150 | tail -12
