"""GPU: the drop-in path end to end - lade.augment_all(); lade.config_lade(...); USE_LADE=1 model.generate(...)
on a HuggingFace LlamaForCausalLM routes into the HIP lookahead loop and reproduces plain HF greedy decoding."""
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture()
def hf_model():
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                      max_position_embeddings=512, rms_norm_eps=1e-6, tie_word_embeddings=False, pad_token_id=0, bos_token_id=1, eos_token_id=None)
    m = LlamaForCausalLM(cfg)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() > 1:
                p.normal_(0, 0.05)
    return m.float().cuda().eval()


def test_generate_with_use_lade_equals_plain_hf_greedy(hf_model, monkeypatch):
    import lade
    from transformers import GenerationMixin
    orig = GenerationMixin._sample
    prompt = torch.tensor([[1, 5, 9, 17, 33, 5, 9, 17, 44, 5, 9]], device="cuda")
    try:
        monkeypatch.delenv("USE_LADE", raising=False)
        plain = hf_model.generate(prompt, attention_mask=torch.ones_like(prompt), do_sample=False, max_new_tokens=40)
        lade.augment_all()
        lade.config_lade(LEVEL=4, WINDOW_SIZE=5, GUESS_SET_SIZE=5, DEBUG=1)
        monkeypatch.setenv("USE_LADE", "0")
        again = hf_model.generate(prompt, attention_mask=torch.ones_like(prompt), do_sample=False, max_new_tokens=40)
        assert torch.equal(plain, again)
        monkeypatch.setenv("USE_LADE", "1")
        random.seed(1)
        out = hf_model.generate(prompt, attention_mask=torch.ones_like(prompt), do_sample=False, max_new_tokens=40)
        assert out.shape == plain.shape and torch.equal(out, plain), (out.tolist(), plain.tolist())
        gen, steps, ratio = lade.decoding.CONFIG_MAP["log"][-1]
        assert gen == 40 and steps <= 40
        random.seed(2)
        smp = hf_model.generate(prompt, attention_mask=torch.ones_like(prompt), do_sample=True, temperature=0.7, top_k=0, top_p=1.0, max_new_tokens=24)
        assert smp.shape[1] == prompt.shape[1] + 24 and int(smp.max()) < 256
    finally:
        GenerationMixin._sample = orig
        lade.decoding.FUNC_MAP.pop("_sample", None)
        lade.decoding.CONFIG_MAP.clear()


def test_jforward_multilevel_boundary_matches_oracle_step(hf_model):
    """The model-step boundary with the reference's signature (lade/models/modeling_llama.py:1381): prefill call,
    a steady-state call with candidates, a lookahead-parallel shard, then the caller-side cache edit - logits against
    the oracle's dense-mask model step on the same weights."""
    import lade
    from oracle import lade_oracle as O
    from lookaheaddecoding_amd import hf
    lade.augment_llama()
    assert type(hf_model).jforward_multilevel is hf.jforward_multilevel
    cfg = hf.config_from_hf(hf_model)
    om = O.OracleLlama(cfg, {k: v.detach().float().cpu() for k, v in hf.weights_from_hf(hf_model).items()})
    W, N = 5, 4
    gs = N - 1
    rng = random.Random(7)
    prompt = [rng.randrange(3, 250) for _ in range(23)]
    L0 = [rng.randrange(3, 250) for _ in range(W + N - 3)]
    dev = "cuda"

    def call(ids, pos, past_tokens, guess, fill_level, pkv):
        return hf_model.jforward_multilevel(input_ids=torch.tensor([ids], device=dev), position_ids=torch.tensor([pos], device=dev),
                                            attention_mask=torch.ones(1, (len(pkv) if pkv is not None else 0) + len(ids), dtype=torch.long, device=dev),
                                            past_key_values=pkv, past_tokens=past_tokens, guess_tokens=guess, return_dict=True, level=N,
                                            WINDOWS_SIZE=W, guess_size=gs, fill_level=fill_level, dist_workers=1, local_rank=0, use_flash=False)

    def check(out, ref, tag):
        assert out.logits is None and out.kvcache_len == ref.kvcache_len, tag
        assert torch.allclose(out.out_logits[0].cpu(), ref.out_logits, atol=2e-4, rtol=1e-4), tag
        assert torch.allclose(out.inp_logits[0].cpu(), ref.inp_logits, atol=2e-4, rtol=1e-4), tag
        if ref.guess_logits is not None:
            assert torch.allclose(out.guess_logits[0].cpu(), ref.guess_logits, atol=2e-4, rtol=1e-4), tag
        else:
            assert out.guess_logits is None, tag

    # 1. prefill: prompt + level 0 (past_tokens[1] is None -> plain causal)
    cache = om.new_cache()
    pt = [list(L0)] + [None] * (N - 2)
    ref = O.model_step(om, cache, prompt, list(range(len(prompt))), pt, None, 0, gs)
    out = call(prompt, list(range(len(prompt))), pt, None, 0, None)
    check(out, ref, "prefill")
    assert len(out.past_key_values) == len(prompt) + len(L0) and out.step_len == len(prompt) + len(L0)
    # the caller keeps only the prompt rows (lade/decoding.py:1130-1137 crops to kvcache_len)
    pkv = out.past_key_values.crop(out.kvcache_len)
    O.kv_truncate(cache, ref.kvcache_len)
    # 2. steady state: full window + 2 candidates, one input token
    lv = [[rng.randrange(3, 250) for _ in range(W - 1)]] + [[rng.randrange(3, 250) for _ in range(W)] for _ in range(N - 2)]
    guess = [rng.randrange(3, 250) for _ in range(2 * gs)]
    nxt, P = 77, len(prompt)
    ref = O.model_step(om, cache, [nxt], [P], lv, guess, N - 2, gs)
    out = call([nxt], [P], lv, guess, N - 2, pkv)
    check(out, ref, "steady")
    assert out.step_len == P + 1 + sum(len(x) for x in lv) + len(guess)
    # caller-side edit: accept candidate 1 with 2 hits -> its rows move down, then crop
    max_hit, idx = 2, 1
    src = out.step_len - len(guess) + idx * gs
    pkv = out.past_key_values.move_rows(src, out.kvcache_len, max_hit).crop(out.kvcache_len + max_hit)
    O.kv_commit(cache, ref.kvcache_len, ref.step_len, len(guess), max_hit, idx, gs)
    # 3. a lookahead-parallel shard (rank 1 of 2: columns 3..5) with re-fed hits as inputs (3 input tokens)
    c0, c1 = 3, 5
    sh = [lv[0][:c1 - 1]] + [l[c0:c1] for l in lv[1:]]
    ins = [rng.randrange(3, 250) for _ in range(3)]
    pos = [P + 3 + i for i in range(3)]
    ref = O.model_step(om, cache, ins, pos, sh, guess[:gs], N - 2, gs)
    out = call(ins, pos, sh, guess[:gs], N - 2, pkv)
    check(out, ref, "lp shard")


def test_streamer_receives_tokens_step_by_step(hf_model, monkeypatch):
    """HF streamers get the accepted tokens of every step as they are accepted (lade/decoding.py:1199-1200), greedy and sampling."""
    import lade
    from transformers import GenerationMixin
    orig = GenerationMixin._sample

    class Collect:
        def __init__(self):
            self.chunks, self.ended = [], False

        def put(self, value):
            self.chunks.append(value.reshape(-1).tolist())

        def end(self):
            self.ended = True

    prompt = torch.tensor([[1, 5, 9, 17, 33, 5, 9, 17, 44, 5, 9]], device="cuda")
    try:
        lade.augment_all()
        lade.config_lade(LEVEL=4, WINDOW_SIZE=5, GUESS_SET_SIZE=5, DEBUG=0)
        monkeypatch.setenv("USE_LADE", "1")
        for kw in (dict(do_sample=False), dict(do_sample=True, temperature=0.7, top_k=0, top_p=1.0)):
            st = Collect()
            random.seed(3)
            out = hf_model.generate(prompt, attention_mask=torch.ones_like(prompt), max_new_tokens=30, streamer=st, **kw)
            assert st.ended and st.chunks[0] == prompt[0].tolist()                 # HF hands the prompt over first
            streamed = [t for c in st.chunks[1:] for t in c]
            assert streamed == out[0, prompt.shape[1]:].tolist()
            assert len(st.chunks) > 3                                               # several steps, not one final dump
    finally:
        GenerationMixin._sample = orig
        lade.decoding.FUNC_MAP.pop("_sample", None)
        lade.decoding.CONFIG_MAP.clear()


def test_tokens_per_s_through_the_generate_surface_equals_the_engine_level_step():
    """the surface north_star says to keep - lade.augment_all(); lade.config_lade(...); USE_LADE=1 model.generate() on a HuggingFace
    LlamaForCausalLM - must cost what the engine-level loop costs (bench.py drives LookaheadDecoder directly; its `via_generate` leg is this measurement at
    the full depth).  A 4-layer model of the 7B width, bf16, W=15 N=5 G=15 as in BASELINE config 2: decode ms per step through generate() against the same
    steps on a LookaheadDecoder over a StepEngine of the same shape."""
    import importlib.util
    import time
    from conftest import ROOT
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    from lookaheaddecoding_amd.weights import make_config, random_weights_torch
    spec = importlib.util.spec_from_file_location("bench_mod_gen", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cfg = make_config("llama2-7b")
    cfg["layers"] = 4
    W, N, G, P, new = 15, 5, 15, 512, 160
    # engine level: the loop bench.py times
    w = random_weights_torch(cfg, seed=0, dtype=torch.bfloat16, device="cuda")
    eng = StepEngine(cfg, w, dtype=torch.bfloat16, device="cuda", max_seq=P + 4 * new + 256, max_T=640, consume_weights=True)
    dec = LookaheadDecoder(eng, W, N, G, use_graph=True)
    prompt = torch.randint(3, cfg["vocab"], (P,), generator=torch.Generator().manual_seed(123)).tolist()
    dec.start(prompt, rng=random.Random(1))
    for _ in range(N - 1 + 16):
        dec.step()
    best = float("inf")
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(48):
            dec.step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 48 * 1e3)
    del dec, eng
    torch.cuda.empty_cache()
    out = bench.via_generate({}, cfg, torch.bfloat16, P, W, N, G, new, best, None)
    assert "error" not in out, out
    assert out["tokens"] == new and out["steps"] >= new - 4 and out["step_compression"] <= 1.05          # random weights accept (almost) nothing
    ratio = out["surface_over_engine_step"]
    assert 0.85 < ratio < 1.15, (ratio, out["decode_ms_per_step"], best)
    assert out["decode_tokens_per_s"] > 2.0 * out["use_lade_0"]["decode_tokens_per_s"]                   # and transformers' own loop is far behind
