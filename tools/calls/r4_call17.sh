#!/bin/bash
# round 4, call 17: step tail - embedding lookup inside the first norm, argmax inside the lm_head GEMM (LADE_FUSE_TAIL).  Kernel and
# engine tests, the end-to-end suites, then c2 with the switch off / on in alternation on one box, then a kernel trace of the fused step
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ktile.py -x -q -k "argmax_epilogue or embed_rmsnorm or fused_tail" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_lp.py tests/test_gpu_hf.py -x -q 2>&1 | tail -6
for rep in 1 2; do
  for v in 0 1; do
    LADE_FUSE_TAIL=$v timeout 600 python bench.py --config c2 --steps 32 --warmup 8 --no-cpu-baseline --blocks 3 2> $OUT/ft_c2_${v}_$rep.err | grep "^{" > $OUT/ft_c2_${v}_$rep.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/ft_c2_${v}_$rep.json"))
    print("c2 fuse=$v rep $rep", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "plain", d["plain_decode"]["ms_per_token"], "hot", d["hot_regime"]["value"], "gpu-only", d["step_gpu_only"]["ms_per_step_back_to_back"])
except Exception as e:
    print("c2 fuse=$v $rep FAILED", e); print(open("$OUT/ft_c2_${v}_$rep.err").read()[-1500:])
PY
  done
done
cd /tmp
rm -rf $OUT/ft_prof
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/ft_prof -o runc -- python $ROOT/bench.py --config c2 --steps 32 --warmup 8 --no-cpu-baseline --no-extras --blocks 1 > $OUT/ft_prof.log 2>&1
cd $ROOT
T=$(find $OUT/ft_prof -name "*kernel_trace.csv" | head -1)
python tools/trace_medians.py $T --steps > $OUT/ft_kernel_medians.txt 2>&1
grep -A16 "per kernel name inside" $OUT/ft_kernel_medians.txt
grep -B2 -A12 "steady step: [0-9]* of" $OUT/ft_kernel_medians.txt | head -30
tail -12 $OUT/ft_kernel_medians.txt | head -0
awk '/^steady step: [0-9]+ of/{f=1} f' $OUT/ft_kernel_medians.txt | grep -B8 "greedy_post_step" | tail -12
rm -rf $OUT/ft_prof
