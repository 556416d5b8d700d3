#!/bin/bash
# round 3, GPU call 8: flakiness of the 2-rank gloo generate() test (faulthandler armed), the LP collective inside the step under rocprofv3
set -u
cd "$(dirname "$0")/../.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do
  timeout 400 python -m pytest tests/test_gpu_bench_pins.py -x -q -k "two_gloo" > $OUT/pytest_gloo_$i.log 2>&1; rc=$?
  echo "gloo run $i rc=$rc $(tail -1 $OUT/pytest_gloo_$i.log)"
  [ $rc -ne 0 ] && { grep -v "^$" $OUT/pytest_gloo_$i.log | tail -120; break; }
done
for coll in torch abi; do
  rm -rf /tmp/lp_$coll
  (cd /tmp && LADE_LP_COLLECTIVE=$coll timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp_$coll -- python $ROOT/bench.py --gpus 1 --force-lp --layers 8 --steps 32 --warmup 8 --blocks 1 --no-cpu-baseline --no-extras > /tmp/lp_$coll.log 2>&1)
  grep "^{" /tmp/lp_$coll.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$coll', d['value'], d['ms_per_step'], d['config']['collective'], d['config']['collective_ranks'])"
  f=$(find /tmp/lp_$coll -name "*kernel_stats.csv" | head -1)
  python tools/kstats.py $f 400 | grep -i "nccl\|rccl\|gather\|lp_pack\|lp_reduce\|build_inputs\|argmax" > $OUT/lp_${coll}_collective_kernels.txt
  cat $OUT/lp_${coll}_collective_kernels.txt
done
