#!/bin/bash
# round 3, GPU call 9: GEMM ring depth A/B (default 3 stages vs up to 5 / 8), split and unsplit, with the engine's own autotune on top
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
L=$ROOT/lookaheaddecoding_amd
for rep in 1 2; do
 for v in "" _ns5 _ns8; do
  echo "== lib [$v] rep $rep"
  LADE_HIP_LIB=$L/liblade_hip$v.so M=60 timeout 120 python tools/gemm_flags.py 2>&1 | grep "sum"
  LADE_HIP_LIB=$L/liblade_hip$v.so M=128 timeout 120 python tools/gemm_flags.py 2>&1 | grep "sum"
  LADE_HIP_LIB=$L/liblade_hip$v.so SHAPES=13b M=120 timeout 120 python tools/gemm_flags.py 2>&1 | grep "sum"
  [ $rep -eq 1 ] && LADE_HIP_LIB=$L/liblade_hip$v.so timeout 200 python tools/gemm_qkv_unsplit_probe.py 2>&1 | grep "qkv-7B.*S=1\|qkv-7B.*S=2 bn=96\|qkv-7B.*S=5 bn=128"
 done
done > $OUT/gemm_ring_ab.txt 2>&1
cat $OUT/gemm_ring_ab.txt
