#!/bin/bash
# round 6, call 13: the lookahead-parallel expectation on the round-6 build (160-row class, two-stage ring, shipped table): every rank's shard of the reference's default
# W=60 N=8 G=60 at R = 1 / 2 / 4 / 8 timed on one GPU (tools/lp_curve.py), + the BASELINE W=15 curve
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6m
{
timeout 1500 python tools/lp_curve.py 7b 60 8 60 2>&1 | grep -v amdgpu.ids
timeout 900 python tools/lp_curve.py 7b 15 5 15 2>&1 | grep -v amdgpu.ids
} | tee gpurun_out/r6m/lp_curve_7b.txt
