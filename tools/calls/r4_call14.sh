#!/bin/bash
# round 4, call 14: what the activation re-reads cost the GEMMs (ablation), 7B at 60 rows and 13B at 120 rows
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
for d in 0 128 1 129 5 133 0 128; do LADE_GEMM_DBG=$d timeout 200 python tools/gemm_ingest_probe.py 2>&1 | grep "LADE_GEMM_DBG" | tee -a $OUT/gemm_ingest.txt | cut -c1-330; done
for d in 0 128 5 133; do MODEL=13b M=120 LADE_GEMM_DBG=$d timeout 200 python tools/gemm_ingest_probe.py 2>&1 | grep "LADE_GEMM_DBG" | tee -a $OUT/gemm_ingest.txt | cut -c1-330; done
