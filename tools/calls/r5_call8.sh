#!/bin/bash
# round 5, call 8 (last): the whole GPU suite + smoke on the final tree, and what the flash_attn_func drop-in costs per call
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python tools/flash_boundary_bench.py 2>&1 | grep -v amdgpu.ids
