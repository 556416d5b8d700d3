#!/bin/bash
# round 6, call 8: the tree with the tuner split out of engine.py (tuning.py), the 64-row GQA attention launch and the eight-unit GEMM build: whole GPU suite, smoke(),
# the 70B rows of the shipped decision table (tools/make_tune_table.py merges them into a copy of the table), the default bench line
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6h
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $OUT/smoke.txt
cp lookaheaddecoding_amd/tuned/gfx950_256cu.json $OUT/tuned.json
timeout 3000 python tools/make_tune_table.py $PWD/$OUT/tuned.json 70b:bf16 2> $OUT/make_tune_table.err | tee $OUT/make_tune_table.txt
tail -2 $OUT/make_tune_table.err | cut -c1-300
timeout 900 python bench.py 2> $OUT/bench_c2.err | grep "^{" > $OUT/bench_c2.json; python -c "
import json; d=json.load(open('$OUT/bench_c2.json')); print(d['value'], d['ms_per_step'], d['parity'], d['via_generate']['decode_tokens_per_s'], d['roofline']['traffic'], d['config']['kernel_decisions'][:70])"
