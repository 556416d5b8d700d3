#!/bin/bash
# round 5, call 6: the whole GPU suite + smoke on the round's final build, then the final evidence set (tools/make_profiles_r5.sh)
set -u
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/make_profiles_r5.sh 2>&1 | tail -40
