#!/bin/bash
# round 6, call 6: the round's evidence set from the build with the shipped decision table (tools/make_profiles_r6.sh: bench lines c2 / c4 / c5 / c3 / f16 / driver form,
# kernel trace, attention counters), the tune-file test on the new header, the GQA attention launch judged by traffic (tools/attn_gqa_sweep.sh)
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6p
timeout 600 python -m pytest tests/test_gpu_tune_file.py -m gpu -q 2>&1 | tail -3
bash tools/make_profiles_r6.sh 2>&1 | tail -40
bash tools/attn_gqa_sweep.sh gpurun_out/r6p/attn_gqa_sweep.txt 2>&1 | tail -12
