#!/bin/bash
# round 3, GPU call 1: XCD census / hand-off probe, the new pin tests, the whole GPU suite, attention A/B + timeline
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 tools/xcd_probe > $OUT/xcd_probe.txt 2>&1; echo "xcd_probe rc=$?"; cat $OUT/xcd_probe.txt
timeout 900 python -m pytest tests/test_gpu_bench_pins.py -x -q -s > $OUT/pytest_pins.log 2>&1; echo "pins rc=$?"; tail -15 $OUT/pytest_pins.log
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "gpu suite rc=$?"; tail -5 $OUT/pytest_gpu.log
L=$ROOT/lookaheaddecoding_amd
for rep in 1 2; do
  for v in "" _noslp; do
    echo "== variant [$v] rep $rep"
    LADE_HIP_LIB=$L/liblade_hip$v.so timeout 120 python tools/attn_bench.py --T 60 120 --P 128 2016 4096 --splits 0 2>&1 | grep "T="
    LADE_HIP_LIB=$L/liblade_hip$v.so timeout 120 python tools/attn_bench.py --T 60 --P 2016 --H 64 --Hkv 8 --splits 0 4 8 2>&1 | grep "T="
  done
done > $OUT/attn_ab1.txt 2>&1
cat $OUT/attn_ab1.txt
LADE_ATTN_DBG=16 LADE_HIP_LIB=$L/liblade_hip_tl.so timeout 120 python tools/attn_bench.py --T 60 --P 128 2016 --splits 0 --reps 50 > $OUT/attn_timeline.txt 2>&1
cat $OUT/attn_timeline.txt
