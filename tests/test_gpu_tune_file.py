"""LADE_TUNE_FILE: the kernel decisions of a model persist across processes and engines, so that every reader of one file launches the same
kernels (same 16-bit rounding, same token stream) and skips the tuning passes.  A table for another GPU / ABI is refused."""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg():
    from lookaheaddecoding_amd.weights import make_config
    return make_config(dict(hidden=1024, inter=2816, layers=2, heads=8, kv_heads=8, head_dim=128, vocab=4096))


def _fresh_process_state():
    """what a new process starts with: no in-process decision caches"""
    from lookaheaddecoding_amd import engine
    engine._TUNE_CACHE.clear()
    engine._TUNE_TIMES.clear()
    engine._TUNE_RANKED.clear()
    engine._STEP_TUNE_CACHE.clear()


def _step(eng, T=60, P=60, seed=5):
    from lookaheaddecoding_amd import ops
    g = torch.Generator(device="cuda").manual_seed(seed)
    ids = torch.randint(3, eng.V, (T,), device="cuda", dtype=torch.int32, generator=g)
    pos = torch.arange(P, P + T, device="cuda", dtype=torch.int32)
    sel = torch.arange(T - 8, T, device="cuda", dtype=torch.int32)
    eng.reset()
    pre = torch.randint(3, eng.V, (P,), device="cuda", dtype=torch.int32, generator=g)
    eng.forward(pre, torch.arange(P, device="cuda", dtype=torch.int32), ops.StepMask(T=P, P=0, is_prefill=True), sel, 0)
    return eng.forward(ids, pos, ops.StepMask(T=T, P=P, is_prefill=True), sel, 8).float().cpu()


def test_tune_file_is_written_adopted_by_a_differently_sized_engine_and_refused_on_mismatch(tmp_path, monkeypatch):
    from lookaheaddecoding_amd import cabi
    from lookaheaddecoding_amd.engine import StepEngine
    from lookaheaddecoding_amd.weights import random_weights_torch
    path = tmp_path / "tune.json"
    monkeypatch.setenv("LADE_TUNE_FILE", str(path))
    cfg = _cfg()
    w = random_weights_torch(cfg, seed=3, dtype=torch.bfloat16)
    _fresh_process_state()
    a = StepEngine(cfg, w, dtype=torch.bfloat16, max_seq=2560, max_T=256)
    assert a.tune_loaded == []
    a.prepare([60])
    logits_a = _step(a)
    doc = json.loads(path.read_text())
    assert doc["header"]["device"] == "gfx950" and doc["header"]["abi"] == cabi.ABI_VERSION
    (key, ent), = doc["models"].items()
    assert set(ent["64"]) >= {"wqkv", "wo", "wgu", "wd", "attn"}
    assert "lm_head" in ent["32"]                      # the 8 logits rows of _step

    # a second "process" with a DIFFERENTLY SIZED engine adopts the table: no tuning, same kernels, same bits
    _fresh_process_state()
    b = StepEngine(cfg, w, dtype=torch.bfloat16, max_seq=4096, max_T=512)
    assert sorted(b.tune_loaded) == [64] and 64 in b._refined
    assert all(b.gemm_cfg[(n, 64)] == a.gemm_cfg[(n, 64)] for n in b.LAYER_GEMMS) and b.gemm_cfg[("lm_head", 32)] == a.gemm_cfg[("lm_head", 32)]
    from lookaheaddecoding_amd import engine
    assert not engine._TUNE_CACHE and not engine._STEP_TUNE_CACHE          # nothing was measured
    assert torch.equal(_step(b), logits_a)
    assert not [k for k in engine._TUNE_CACHE if k[2] == 64]                 # ... not by the step either

    # the file decides, not the box: an edited decision is what the next engine launches
    forced = [2, 64, 2, 1, 1, 0]
    assert list(a.gemm_cfg[("wo", 64)] or ()) != forced
    doc["models"][key]["64"]["wo"] = forced
    path.write_text(json.dumps(doc))
    _fresh_process_state()
    c = StepEngine(cfg, w, dtype=torch.bfloat16, max_seq=2560, max_T=256)
    assert c.gemm_cfg[("wo", 64)] == tuple(forced)
    d = StepEngine(cfg, w, dtype=torch.bfloat16, max_seq=3072, max_T=320)
    assert torch.equal(_step(c), _step(d))               # two readers of one file: one stream
    ref = logits_a
    assert torch.allclose(_step(c), ref, atol=0.15, rtol=0.05)             # (another split count: same logits up to bf16 rounding)

    # a table of another GPU / library is refused, loudly
    doc["header"]["device"] = "some other accelerator"
    path.write_text(json.dumps(doc))
    with pytest.raises(cabi.LadeHipError, match="refusing to adopt"):
        StepEngine(cfg, w, dtype=torch.bfloat16, max_seq=1024, max_T=256)
