"""CPU: the bench line committed under profiles/ carries every field of the bench.py contract (a guard against silently
dropping one when bench.py changes; the line itself is produced on the GPU box)."""
import json
import os

from conftest import ROOT


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r2_bench.json")) as f:
        line = [l for l in f.read().splitlines() if l.strip().startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["unit"] == "tokens/s" and d["higher_is_better"] is True and d["n_gpus"] == 1 and d["dtype"] in ("bf16", "f16")
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] >= r["algorithmic_bytes"] * 0.9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    assert abs(d["value"] - d["step_compression"] * 1e3 / d["ms_per_step"]) / d["value"] < 0.02      # tokens/s = S / step time
