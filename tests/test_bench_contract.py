"""CPU: the bench line committed under profiles/ carries every field of the bench.py contract (a guard against silently
dropping one when bench.py changes; the line itself is produced on the GPU box)."""
import json
import os

from conftest import ROOT


def test_committed_bench_line_has_the_contract_fields():
    with open(os.path.join(ROOT, "profiles", "r6_bench.json")) as f:
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
    # round 6: the counter traffic is there on every box (frozen attention launch: 6 splits at config 2; else the nearest profiled split count, marked ESTIMATE)
    assert r["traffic"] is not None and r["algorithmic_bytes"] * 0.9 <= r["traffic"] <= r["algorithmic_bytes"] * 1.5 and "profiles/" in r["traffic_source"]
    assert r["launch_parameters"]["n_splits"] == 6 and r["launch_parameters"]["wg_rows"] == 128 and r["launch_parameters"]["split_mode"] == 0
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    assert abs(d["value"] - d["step_compression"] * 1e3 / d["ms_per_step"]) / d["value"] < 0.02      # tokens/s = S / step time
    # round 4: the port's speed calibrated against the shim-loaded reference (oracle/cpu_calibration.json, written in the build container)
    cal = c["calibration_vs_reference"]
    assert cal and all(0.5 < x < 1.5 for x in cal["port_over_reference_time"]) and "calibration" in c["sample"]
    with open(os.path.join(ROOT, "oracle", "cpu_calibration.json")) as f:
        assert [x["port_over_reference_time"] for x in json.load(f)["cases"]] == cal["port_over_reference_time"]
    # the regime the reference publishes (S ~ 1.6-2.3): round 5 reports the in-range trial of a FIXED search itself, and what does
    # not depend on how often this random model accepts - the speed-up at the published S values and the break-even S
    m = d["mid_regime"]
    assert m["unit"] == "tokens/s" and abs(m["speedup_vs_plain"] - m["step_compression"] * m["plain_ms_per_token"] / m["ms_per_step"]) < 0.01
    assert m["status"] == "in_range" and m["in_published_range"] is True and 1.6 <= m["step_compression"] <= 2.3
    assert m["scales_tried_S"][0][0] == 24.0 and [s_ for s_, _ in m["scales_tried_S"][:3]] == [24.0, 28.79, 34.54]          # the fixed grid, in order
    sp_ = m["speedup_at_published_S"]
    assert abs(sp_["1.95"] - 1.95 * m["plain_ms_per_token"] / m["ms_per_step"]) < 0.01 and sp_["1.6"] < sp_["1.95"] < sp_["2.3"]
    assert abs(m["break_even_S"] - m["ms_per_step"] / m["plain_ms_per_token"]) < 0.01
    # every emitted token against the plain one-token step on its own prefix: the plain argmax, or within a few spacings of the dtype
    for k in ("mid_regime", "hot_regime"):
        gc_ = d[k]["greedy_check"]
        assert gc_["tokens"] == 64 and gc_["plain_argmax_of_own_prefix"] >= 60 and gc_["in_dtype_spacings"] <= 4.0, (k, gc_)
    # ... and where the driver's parser keeps it: config.parity, config.spread_ms_per_step_blocks, the calibration factor beside cpu_baseline.value
    assert "tests/test_gpu_e2e.py" in d["config"]["parity"]["bit_identical_greedy_ids"] and "of 64" in d["config"]["parity"]["this_run"]["mid_regime_teacher_forced"]
    assert d["config"]["spread_ms_per_step_blocks"] == d["spread"]["ms_per_step_blocks"]
    assert abs(c["reference_equivalent_value"] - c["value"] * c["port_over_reference_time"]) < 1e-3 and c["port_over_reference_time"] == max(cal["port_over_reference_time"])
    # the blocks after the contract's: rows fed per step and the slowest step - no collector stall (a 37-44 ms step) any more
    sp = d["spread"]
    assert len(sp["rows_per_step_blocks"]) == sp["blocks"] and all(r >= 60.0 for r in sp["rows_per_step_blocks"])
    assert sp["slowest_step_ms_and_index_blocks"][0] is None and all(m < 2.0 * d["ms_per_step"] for m, _i in sp["slowest_step_ms_and_index_blocks"][1:])
    # the host's turn-around from ALTERNATING blocks (real loop / back-to-back replays): within the block-to-block noise, <= ~2 % of a step
    g = d["step_gpu_only"]
    assert g["valid"] and len(g["pairwise_differences_us"]) >= 2 and abs(g["host_turnaround_us_per_step"]) < 0.02 * d["ms_per_step"] * 1e3
    # the attention launch as the in-step tuner decided it, and the pair's time with and without RoPE / append
    r5 = d["roofline"]
    lp_ = r5["launch_parameters"]
    assert lp_["wg_rows"] in (32, 64, 128) and lp_["n_splits"] >= 2 and lp_["rope_kv_append_fused_into_the_launch"] in (False, True)
    assert r5["launch_us_with_rope_append"] > r5["launch_us"] > 0
    # round 6: the metric's second half and the surface's speed where the driver's parser keeps them (config), the surface leg itself, the decision table
    cfg = d["config"]
    assert "step-compression" in d["metric"] and cfg["step_compression"] == d["step_compression"]
    assert abs(cfg["lookahead_over_plain_step"] - d["ms_per_step"] / d["plain_decode"]["ms_per_token"]) < 0.01
    assert cfg["mid_regime"]["break_even_S"] == m["break_even_S"] and cfg["mid_regime"]["speedup_at_published_S"] == m["speedup_at_published_S"]
    assert "tuned/gfx950_256cu.json" in cfg["kernel_decisions"]
    v = d["via_generate"]
    assert "USE_LADE=1 model.generate" in v["surface"] and "lade.augment_all()" in v["surface"] and "error" not in v
    assert v["tokens"] == 256 and v["steps"] >= 250 and abs(v["surface_over_engine_step"] - 1.0) < 0.03          # within 3 % of the engine-level step
    assert abs(v["decode_tokens_per_s"] - (v["tokens"] - 1) / (v["seconds"] - v["one_token_call_ms"] * 1e-3)) / v["decode_tokens_per_s"] < 0.01
    assert v["use_lade_0"]["decode_tokens_per_s"] < v["decode_tokens_per_s"] and cfg["via_generate"]["decode_tokens_per_s"] == v["decode_tokens_per_s"]
    if "parity" in d:          # (lines written by the final bench.py of the round)
        assert "fp32 engine only" in d["parity"]["bit_identical_to_reference"] and d["parity"]["kernel_decisions_box_independent"] is True


def test_step_stream_bytes_model_and_the_committed_figure():
    """bench.py's `step_stream`: the bytes a decode step cannot avoid reading.  The byte model against the 7B shape by hand, and the
    committed line consistent with it (achieved = bytes / ms_per_step, frac = achieved / peak)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cfg = dict(hidden=4096, inter=11008, layers=32, heads=32, kv_heads=32, head_dim=128, vocab=32000)
    per_layer = (3 * 4096 * 4096 + 4096 * 4096 + 3 * 11008 * 4096) * 2                   # qkv, o, gate / up / down in bf16
    kv = 2 * 32 * 2091 * 128 * 2
    assert bench.step_stream_bytes(cfg, 2091, 1) == 32 * (per_layer + kv) + 32000 * 4096 * 2
    assert bench.step_stream_bytes(cfg, 2091, 0) == 32 * (per_layer + kv)
    with open(os.path.join(ROOT, "profiles", "r6_bench.json")) as f:
        d = json.loads([l for l in f.read().splitlines() if l.strip().startswith("{")][-1])
    s = d["step_stream"]
    assert s["bound"] == "hbm" and s["unit"] == "GB/s" and s["peak"] == 8000.0
    assert abs(s["achieved"] - s["bytes_per_step"] / (d["ms_per_step"] * 1e-3) / 1e9) / s["achieved"] < 0.01
    assert abs(s["frac"] - s["achieved"] / s["peak"]) < 1e-3 and 0.0 < s["frac"] < 1.0


def test_projections_object_of_the_committed_line_is_consistent():
    """bench.py's `projections` (what the engine's autotune timed for the kernels it chose): weight bytes of the 7B shape, TB/s = bytes / time,
    the layer sum, and the layout the line says the GEMMs stream"""
    # (a line whose engine TUNED IN ITS PROCESS: a line that runs on the shipped decision table has no timings of its own to report)
    with open(os.path.join(ROOT, "profiles", "r6_bench_call4.json")) as f:
        d = json.loads([l for l in f.read().splitlines() if l.strip().startswith("{")][-1])
    p = d["projections"]
    assert p["row_class"] == 64 and p["weight_layout"] == "k-tile-major" and "K-tile-major" in d["config"]["weight_layout"]
    want_mb = {"wqkv": 3 * 4096 * 4096 * 2 / 1e6, "wo": 4096 * 4096 * 2 / 1e6, "wgu": 2 * 11008 * 4096 * 2 / 1e6, "wd": 11008 * 4096 * 2 / 1e6}
    tot = 0.0
    for n, mb in want_mb.items():
        e = p[n]
        assert abs(e["weight_mb"] - mb) < 0.1 and abs(e["tb_per_s"] - mb / e["us"]) < 0.02 and abs(e["frac_of_8_tb_per_s"] - e["tb_per_s"] / 8.0) < 2e-3
        assert e["kernel"] == "library" or len(e["kernel"]) == 6          # (mb, bn, n_split, mt, nt, ring)
        tot += e["us"]
    assert abs(p["layer_sum_us"] - tot) < 0.05 and 0.3 < p["layer_tb_per_s"] / 8.0 < 1.0
    # the decisions re-taken inside a step: per projection both choices and their per-layer times in the 8-layer probe
    t = {k: v for k, v in p["in_step_tuning"].items() if k != "attn"}
    assert set(t) == set(want_mb)
    for n, e in t.items():
        assert e["ms_per_layer_in_step_choice"] <= e["ms_per_layer_isolated_choice"] + 1e-9 and e["candidates"] >= 2
        assert e["in_step_choice"] == p[n]["kernel"]
