#!/bin/bash
# K-tile-major weight layout probe
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
timeout 400 python tools/gemm_ktile_probe.py 7b 60 30 120 > gpurun_out/r3/gemm_ktile_probe.txt 2>&1; echo "probe rc=$?"
timeout 200 python tools/gemm_ktile_probe.py 13b 120 >> gpurun_out/r3/gemm_ktile_probe.txt 2>&1; echo "probe13 rc=$?"
grep -v "^      " gpurun_out/r3/gemm_ktile_probe.txt | tail -30
