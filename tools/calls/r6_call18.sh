#!/bin/bash
# round 6, call 18: the attention K / V stream non-temporal + sc1 (new default) against the default policy of rounds 1-5 (LADE_DEBUG=attn_dbg=128), second box: c2 / c4 / c5
# alternating; then the attention tests and the whole GPU suite on the new default
set -u
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r6s
mkdir -p $OUT
for rep in 1 2; do
  for arm in ntsc1 old; do
    dbg=""; [ $arm = old ] && dbg="attn_dbg=128"
    for c in c2 c4; do
      LADE_DEBUG=$dbg timeout 1200 python bench.py --config $c --no-cpu-baseline --no-generate --blocks 3 2> $OUT/${c}_${arm}_$rep.err | grep "^{" > $OUT/${c}_${arm}_$rep.json
      python - <<PY
import json
try:
    d=json.load(open("$OUT/${c}_${arm}_$rep.json"))
    r=d["roofline"]
    print("$c $arm rep $rep cold", d["value"], d["ms_per_step"], d["spread"]["ms_per_step_blocks"], "| pair us", r["launch_us"], "isolated", r["launch_us_isolated"], "frac", r["frac"], "| plain", d["plain_decode"]["ms_per_token"], "| mid ms", d["mid_regime"]["ms_per_step"], "T", d["mid_regime"]["tokens_per_step_T"], "| hot ms", d["hot_regime"]["ms_per_step"])
except Exception as e:
    print("$c $arm $rep FAILED", e); print(open("$OUT/${c}_${arm}_$rep.err").read()[-600:])
PY
    done
  done
done | tee $OUT/attn_kv_cache_policy_ab2.txt
for arm in ntsc1 old; do
  dbg=""; [ $arm = old ] && dbg="attn_dbg=128"
  LADE_DEBUG=$dbg timeout 1500 python bench.py --config c5 --no-cpu-baseline --no-extras --blocks 3 2> $OUT/c5_${arm}.err | grep "^{" > $OUT/c5_${arm}.json
  python -c "
import json; d=json.load(open('$OUT/c5_${arm}.json')); print('c5 $arm', d['value'], d['ms_per_step'], d['spread']['ms_per_step_blocks'], 'pair', d['roofline']['launch_us'])"
done | tee -a $OUT/attn_kv_cache_policy_ab2.txt
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
