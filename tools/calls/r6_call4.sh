#!/bin/bash
# round 6, call 4: new tests (160-row class bit-identity, tune file, poll fallback, experimental refusal), zero padding rows against the power cap,
# c2 bench with the via_generate leg, c4 with / without the 160-row class in alternation, 13B rows curve with / without it
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6d
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_ktile.py tests/test_gpu_tune_file.py tests/test_gpu_poll.py tests/test_gpu_fused_rope.py tests/test_gpu_flash_boundary.py -m gpu -q 2>&1 | tail -25 | tee $OUT/pytest_new.txt
{
CASES=76,92,120,132,150 LADE_DEBUG=gemm_dbg=0 timeout 300 python tools/clock_probe.py 2>&1 | grep -v amdgpu.ids
CASES=76,92,120,132,150 LADE_DEBUG=gemm_dbg=256 timeout 300 python tools/clock_probe.py 2>&1 | grep -v amdgpu.ids
} | tee $OUT/zero_pad_probe.txt
timeout 900 python bench.py 2> $OUT/bench_c2.err | grep "^{" > $OUT/bench_c2.json; tail -c 1500 $OUT/bench_c2.json; tail -5 $OUT/bench_c2.err
for rep in 1 2; do
  for cls in r6 r5; do
    dbg=""; [ $cls = r5 ] && dbg="row_classes=r5"
    LADE_DEBUG=$dbg timeout 1200 python bench.py --config c4 --no-cpu-baseline --no-generate --blocks 2 2> $OUT/bench_c4_${cls}_$rep.err | grep "^{" > $OUT/bench_c4_${cls}_$rep.json
    python - <<PY
import json
try:
    d=json.load(open("$OUT/bench_c4_${cls}_$rep.json"))
    m,h,f=d["mid_regime"],d["hot_regime"],d["hot_regime_forced"]
    print("c4 classes=$cls rep $rep cold", d["value"], d["ms_per_step"], "| plain", d["plain_decode"]["ms_per_token"], "| mid S", m["step_compression"], "T", m["tokens_per_step_T"], "ms", m["ms_per_step"], "x plain", m["speedup_vs_plain"], "at 1.6/1.95/2.3:", m["speedup_at_published_S"], "| hot S", h["step_compression"], "T", h["tokens_per_step_T"], "ms", h["ms_per_step"], h["value"], "| forced T", f["tokens_per_step_T"], "ms", f["ms_per_step"], f["value"])
except Exception as e:
    print("$cls $rep FAILED", e); print(open("$OUT/bench_c4_${cls}_$rep.err").read()[-800:])
PY
  done
done | tee $OUT/c4_row_classes_ab.txt
{
timeout 900 python tools/rows_curve.py 13b 1 60 96 120 128 132 144 150 156 160 180 192 240 2>&1 | grep -v amdgpu.ids
LADE_DEBUG=row_classes=r5 timeout 900 python tools/rows_curve.py 13b 1 128 132 144 156 160 192 2>&1 | grep -v amdgpu.ids
} | tee $OUT/rows_curve_13b.txt
