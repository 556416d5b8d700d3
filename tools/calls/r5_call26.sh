#!/bin/bash
# round 5, call 26: the step time follows the capacity of the K / V cache that bench.py allocates (calls 23 / 25: 3712 rows 3.81-3.82 ms, 4672: 3.83, 5504: 3.85): confirm on the
# current tree with the capacity as the only variable, then kernel traces at the two ends to see which launch it is
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r5q
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
  for ms in 3652 5492 8192 12288; do
    LADE_BENCH_MAX_SEQ=$ms timeout 600 python bench.py --config c2 --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/cap_${ms}_$rep.err | grep "^{" > $OUT/cap_${ms}_$rep.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/cap_${ms}_$rep.json"))
    print("c2 cache capacity $ms rep $rep", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"])
except Exception as e:
    print("$ms $rep FAILED", e); print(open("$OUT/cap_${ms}_$rep.err").read()[-800:])
PY
  done
done
for ms in 3652 12288; do
  rm -rf /tmp/kt_$ms
  (cd /tmp && LADE_BENCH_MAX_SEQ=$ms timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$ms -- python $ROOT/bench.py --steps 32 --warmup 8 --blocks 1 --no-cpu-baseline --no-extras > /tmp/kt_$ms.log 2>&1)
  python tools/trace_medians.py $(find /tmp/kt_$ms -name "*kernel_trace.csv" | head -1) --steps > $OUT/trace_cap_$ms.txt
  echo "== capacity $ms"; grep "steady step: first" $OUT/trace_cap_$ms.txt; grep -A9 "per kernel name inside" $OUT/trace_cap_$ms.txt | cut -c1-110
done
