#!/bin/bash
# round 6, call 23: the driver's own sequence on the final library - pytest -m gpu, smoke(), python bench.py
set -u
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6w
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r6w/pytest_gpu.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r6w/smoke.txt
s=$(date +%s); timeout 900 python bench.py > gpurun_out/r6w/bench_stdout.txt 2> gpurun_out/r6w/bench.err; echo "bench.py wall $(( $(date +%s) - s )) s rc=$?" | tee gpurun_out/r6w/bench_wall.txt
tail -1 gpurun_out/r6w/bench_stdout.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['via_generate']['decode_tokens_per_s'], d['config']['kernel_decisions'][:50])"
