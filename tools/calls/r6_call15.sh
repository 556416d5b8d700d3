#!/bin/bash
# round 6, call 15: cache-policy bits of the weight stream beyond nt (LADE_DEBUG=gemm_dbg=256: nt + sc1; 512: nt + sc0 + sc1) - time, power, clock of the 13B gate/up launch
# at 60 / 120 / 150 rows (tools/clock_probe.py CASES)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6o
for d in 0 256 512 0 256; do
  CASES=60,120,150 LADE_DEBUG=gemm_dbg=$d timeout 300 python tools/clock_probe.py 2>&1 | grep -v "amdgpu.ids\|idle"
done | tee gpurun_out/r6o/weight_cache_policy_probe.txt
