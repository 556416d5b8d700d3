#!/bin/bash
# round 5, call 2: the whole GPU suite on the round's build (flash_attn_func boundary, fused-RoPE launch parameter, attention launch
# tuned in the step, NaN rule of the argmax kernels, dynamic-NTK state across calls, 8-rank shared-GPU bench), then the evidence set
# (tools/make_profiles_r5.sh: bench lines c2 / c4 / c5 / c3 / f16, the driver's own invocation, kernel trace, PMC passes of the attention pair)
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25
bash tools/make_profiles_r5.sh 2>&1 | tail -60
