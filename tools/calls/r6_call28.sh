#!/bin/bash
# round 6, call 28: the in-tree library once more - smoke(), the kernel parity tests, python bench.py
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6ab
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r6ab/smoke.txt
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ktile.py tests/test_gpu_e2e.py -q -m gpu 2>&1 | grep -E "passed|failed|error" | tee gpurun_out/r6ab/pytest.txt
timeout 900 python bench.py 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['via_generate']['decode_tokens_per_s'])" | tee gpurun_out/r6ab/bench.txt
