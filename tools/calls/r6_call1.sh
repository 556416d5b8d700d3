#!/bin/bash
# round 6, call 1: the register-resident-activation GEMM (csrc/gemm_ra.hpp) against the LDS-ring kernel, isolated, at the BASELINE widths;
# what clock / power the chip sustains under the 64- and 128-row classes; can the lease be switched to CPX partitions
# ? - refused, see below
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6a
{
MODEL=13b M=120 timeout 600 python tools/gemm_ra_probe.py 2>&1 | tail -60
MODEL=7b M=60 timeout 600 python tools/gemm_ra_probe.py 2>&1 | tail -60
MODEL=7b M=120 timeout 600 python tools/gemm_ra_probe.py 2>&1 | tail -60
MODEL=7b M=92 timeout 600 python tools/gemm_ra_probe.py 2>&1 | tail -60
} | tee gpurun_out/r6a/gemm_ra_probe.txt
timeout 300 python tools/clock_probe.py 2>&1 | tee gpurun_out/r6a/clock_probe.txt
# (the partition switch this call first carried was refused by the pool before anything ran: profiles/r6_cpx_rccl.txt)
timeout 30 rocm-smi --showcomputepartition --showmemorypartition 2>&1 | tail -12 | tee gpurun_out/r6a/partition_mode.txt
