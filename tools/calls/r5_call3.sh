#!/bin/bash
# round 5, call 3: the evidence set again on the final bench.py (mid regime: fixed search, the in-range trial itself is reported; host
# turn-around from 5 alternating block pairs; attention tuner: a challenger must win by 1 %), + the bench's own end-to-end tests
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
bash tools/make_profiles_r5.sh 2>&1 | tail -45
timeout 900 python -m pytest tests/test_gpu_examples.py -x -q 2>&1 | tail -5
