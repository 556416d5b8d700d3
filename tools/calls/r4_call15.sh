#!/bin/bash
# round 4, call 15: hunt the start-up stall after an RCCL communicator teardown (tools/lp_stall_repro.py)
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1500 python tools/lp_stall_repro.py 14 > gpurun_out/r4/lp_stall.txt 2>&1; echo "rc=$?"; tail -25 gpurun_out/r4/lp_stall.txt | cut -c1-300
