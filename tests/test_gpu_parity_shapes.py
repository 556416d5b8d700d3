"""Parity at the shapes the bench and BASELINE.json actually use (round-2 additions):

* the chunked prefill (prompt longer than the engine's step width) through every caller: greedy (eager + hipGraph),
  sampling, lookahead-parallel ranks and hf.jforward_multilevel - tokens / steps / logits against the CPU oracle in fp32;
* 7B / 13B / 70B WIDTHS with real random weights (attention and MLP live, bf16, MFMA path): the lookahead stream is the
  plain greedy stream of the same engine, every emitted token is within the stated logit margin of the fp32 oracle, and
  the K/V rows a lookahead run leaves in the cache (appended, committed after hits) equal the rows a plain causal
  prefill of the same tokens writes;
* lookahead parallelism in bf16 on the HIP ranks (R = 2, 8).

Reference: lade/decoding.py:697-1259 (greedy), :137-692 (sampling), :956-1107 (LP), lade/models/modeling_llama.py:124-130
(prefill mask), :1381-1608 (model step)."""
import random
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import lade_oracle as O
from lookaheaddecoding_amd.weights import make_config, random_weights_numpy, random_weights_torch


def _tiny(name, seed, std, dtype, max_seq, max_T):
    from lookaheaddecoding_amd.engine import StepEngine
    cfg = make_config(name, max_pos=max(max_seq, 512))
    w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=seed, std=std).items()}
    return cfg, w, StepEngine(cfg, w, dtype=dtype, max_seq=max_seq, max_T=max_T)


def _long_prompt(n, vocab, seed=11):
    rng = random.Random(seed)
    base = [rng.randrange(3, vocab) for _ in range(23)]
    return [base[i % len(base)] if (i // 7) % 3 else rng.randrange(3, vocab) for i in range(n)]


# ---- (a) chunked prefill -----------------------------------------------------------------------------------------

def test_chunked_prefill_greedy_eager_and_graph_vs_oracle_fp32():
    """prompt 300 + window > max_T = 64: five causal chunks fill the cache before the logits chunk (bench.py's 2048-token
    prompt takes the same path).  Tokens, step count and per-step acceptance are the oracle's."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    cfg, w, eng = _tiny("tiny-d64", 1, 0.05, torch.float32, max_seq=640, max_T=64)
    model = O.OracleLlama(cfg, w)
    prompt = _long_prompt(300, cfg["vocab"])
    for (W, N, G, pfp) in ((5, 3, 3, False), (7, 4, 7, True)):
        ref = O.lookahead_greedy(model, prompt, W, N, G, len(prompt) + 40, random.Random(4), pool_from_prompt=pfp, keep_trace=True)
        for use_graph in (False, True):
            dec = LookaheadDecoder(eng, W, N, G, pool_from_prompt=pfp, use_graph=use_graph)
            out = dec.greedy(prompt, len(prompt) + 40, rng=random.Random(4), keep_trace=True)
            assert out.tokens == ref.tokens and out.steps == ref.steps, (W, N, G, use_graph)
            assert [t["max_hit"] for t in out.trace] == [t.max_hit for t in ref.trace]
            # the first step is the LAST chunk: T <= max_T rows on top of the rows the earlier chunks cached
            assert out.trace[0]["T"] <= 64 and out.trace[0]["P_before"] == len(prompt) + W + N - 3 - out.trace[0]["T"]
            assert out.trace[1]["P_before"] == len(prompt)
    assert eng.plain_greedy(prompt, len(prompt) + 40) == O.plain_greedy(model, prompt, len(prompt) + 40)


def test_chunked_prefill_sampling_vs_oracle_fp32():
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.sampling import make_warper
    cfg, w, eng = _tiny("tiny-d64", 1, 0.05, torch.float32, max_seq=640, max_T=64)
    model = O.OracleLlama(cfg, w)
    prompt = _long_prompt(300, cfg["vocab"])
    warp = dict(temperature=0.7, top_k=0, top_p=1.0)
    ref = O.lookahead_sample(model, prompt, 5, 4, 5, len(prompt) + 32, random.Random(9), torch.Generator().manual_seed(9), pool_from_prompt=True, **warp)
    for use_graph in (False, True):
        dec = LookaheadDecoder(eng, 5, 4, 5, pool_from_prompt=True, use_graph=use_graph)
        out = dec.sample(prompt, len(prompt) + 32, warp=make_warper(**warp), rng=random.Random(9), torch_gen=torch.Generator().manual_seed(9))
        assert out.tokens == ref.tokens and out.steps == ref.steps, use_graph


def test_chunked_prefill_through_jforward_multilevel_vs_oracle():
    """hf.jforward_multilevel with a 700-token prompt (> the 512-row step width of the engine it builds)."""
    import lade
    from transformers import LlamaConfig, LlamaForCausalLM
    from lookaheaddecoding_amd import hf
    torch.manual_seed(0)
    c = LlamaConfig(vocab_size=256, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                    max_position_embeddings=2048, rms_norm_eps=1e-6, tie_word_embeddings=False, pad_token_id=0, bos_token_id=1, eos_token_id=None)
    m = LlamaForCausalLM(c)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() > 1:
                p.normal_(0, 0.05)
    m = m.float().cuda().eval()
    lade.augment_llama()
    cfg = hf.config_from_hf(m)
    om = O.OracleLlama(cfg, {k: v.detach().float().cpu() for k, v in hf.weights_from_hf(m).items()})
    W, N = 5, 4
    prompt = _long_prompt(700, 250)
    L0 = [3 + (i * 13) % 200 for i in range(W + N - 3)]
    pt = [list(L0)] + [None] * (N - 2)
    ref = O.model_step(om, om.new_cache(), prompt, list(range(len(prompt))), pt, None, 0, N - 1)
    out = m.jforward_multilevel(input_ids=torch.tensor([prompt], device="cuda"), position_ids=torch.arange(len(prompt), device="cuda")[None],
                                attention_mask=torch.ones(1, len(prompt), dtype=torch.long, device="cuda"), past_key_values=None, past_tokens=pt,
                                guess_tokens=None, return_dict=True, level=N, WINDOWS_SIZE=W, guess_size=N - 1, fill_level=0, dist_workers=1,
                                local_rank=0, use_flash=False)
    assert out.kvcache_len == ref.kvcache_len == len(prompt) and len(out.past_key_values) == len(prompt) + len(L0)
    assert torch.allclose(out.out_logits[0].cpu(), ref.out_logits, atol=3e-4, rtol=1e-4)
    assert torch.allclose(out.inp_logits[0].cpu(), ref.inp_logits, atol=3e-4, rtol=1e-4)
    # a later step whose cache length exceeds the engine's first allocation grows the cache in place and keeps the rows
    eng = getattr(m, hf._ENGINE_ATTR)
    k_before = eng.k_cache(1)[:, :len(prompt)].clone()
    eng.grow(eng.S_max + 1024, eng.max_T, keep_rows=len(prompt))
    assert torch.equal(eng.k_cache(1)[:, :len(prompt)], k_before) and out.past_key_values.engine is eng


class _ThreadExchange:
    def __init__(self, R):
        self.R, self.bar, self.slots = R, threading.Barrier(R), [None] * R

    def all_gather(self, rank, out, inp):
        torch.cuda.current_stream().synchronize()
        self.slots[rank] = inp.clone()
        self.bar.wait()
        out.copy_(torch.cat([s.to(out.device) for s in self.slots]))
        torch.cuda.current_stream().synchronize()
        self.bar.wait()

    def broadcast(self, rank, t):
        if rank == 0:
            self.slots[0] = t.clone()
        self.bar.wait()
        t.copy_(self.slots[0])
        self.bar.wait()


def _lp_threads(R, make_dec, prompt, max_length, seed):
    """R lookahead-parallel ranks as R threads on cuda:0 (own engine / window / pool / cache each), the collectives replaced
    by an in-process exchange; rank-local kernels and the orchestration are the product's."""
    from lookaheaddecoding_amd import parallel
    torch.zeros(1, device="cuda")
    ex = _ThreadExchange(R)
    results, errors = {}, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            dec = make_dec(rank, parallel.LPContext(rank=rank, world=R))
            be = parallel.HipLPBackend(dec)
            be.broadcast_window = lambda w0, lp: (lambda t: (ex.broadcast(rank, t), t.tolist())[1])(torch.tensor(w0, dtype=torch.int32, device="cuda"))
            with torch.cuda.stream(torch.cuda.Stream()):
                out = parallel.greedy_lp(dec, prompt, max_length, rng=random.Random(seed + 1000 * rank), backend=be, keep_trace=True,
                                         all_gather=lambda o, i: ex.all_gather(rank, o, i))
            results[rank] = out
        except Exception:  # pragma: no cover
            import traceback
            errors.append((rank, traceback.format_exc()))
            try:
                ex.bar.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=run, args=(r,)) for r in range(R)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors[0][1]
    return [results[r] for r in range(R)]


def test_chunked_prefill_lookahead_parallel_ranks_vs_oracle_fp32():
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    cfg = make_config("tiny-d64", max_pos=640)
    w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=1, std=0.05).items()}
    model = O.OracleLlama(cfg, w)
    prompt = _long_prompt(300, cfg["vocab"])
    W, N, G, R = 7, 4, 7, 2
    ref = O.lookahead_greedy(model, prompt, W, N, G, len(prompt) + 32, random.Random(6), pool_from_prompt=True, R=R, keep_trace=False)

    def make_dec(rank, lp):
        eng = StepEngine(cfg, w, dtype=torch.float32, max_seq=640, max_T=64)
        return LookaheadDecoder(eng, W, N, G, lp=lp, pool_from_prompt=True)

    outs = _lp_threads(R, make_dec, prompt, len(prompt) + 32, seed=6)
    for out in outs:
        assert out.tokens == ref.tokens and out.steps == ref.steps


# ---- (d) lookahead parallelism on the MFMA (bf16) path ---------------------------------------------------------------

@pytest.mark.parametrize("R", [2, 8])
def test_lookahead_parallel_bf16_ranks_agree_and_match_single_gpu(R):
    """bf16 end to end under LP: every rank ends with the same token stream (decisions are taken on the gathered records
    only), the stream is the single-GPU plain greedy stream or differs from it only at oracle-valid near-ties, and the
    pool does hit (hits are re-fed: lade/decoding.py:1148-1153)."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    cfg = make_config("tiny-d128", max_pos=512)
    wn = random_weights_numpy(cfg, seed=2, std=0.05)
    w = {k: torch.as_tensor(v) for k, v in wn.items()}
    prompt = [1, 5, 9, 17, 33, 5, 9, 17, 44, 5, 9, 17, 33, 5, 9]
    W, N, G = 15, 5, 15
    n_new = 64

    def make_dec(rank, lp):
        eng = StepEngine(cfg, w, dtype=torch.bfloat16, max_seq=512, max_T=320)
        return LookaheadDecoder(eng, W, N, G, lp=lp, pool_from_prompt=True)

    outs = _lp_threads(R, make_dec, prompt, len(prompt) + n_new, seed=3)
    for out in outs[1:]:
        assert out.tokens == outs[0].tokens and out.steps == outs[0].steps
    eng = StepEngine(cfg, w, dtype=torch.bfloat16, max_seq=512, max_T=320)
    plain = eng.plain_greedy(prompt, len(prompt) + n_new)
    ok, worst = _oracle_margin(cfg, w, torch.bfloat16, outs[0].tokens, len(prompt), tol=0.06)
    assert ok, worst
    if outs[0].tokens != plain:
        assert _oracle_margin(cfg, w, torch.bfloat16, plain, len(prompt), tol=0.06)[0]
    assert outs[0].steps < outs[0].generated, "no n-gram was accepted under LP"


# ---- (b) BASELINE widths with real random weights ------------------------------------------------------------------

def _oracle_margin(cfg, w_cpu_f32, dtype, tokens, n_prompt, tol, rel=0.0):
    """every generated token's fp32-oracle logit is within tol (+ rel * |best logit|) of the oracle's best logit for its prefix
    (oracle = fp32 math on the weights rounded to `dtype`)"""
    wq = {k: v.to(dtype).float() for k, v in w_cpu_f32.items()}
    model = O.OracleLlama(cfg, wq)
    T = len(tokens) - 1
    vis = np.tril(np.ones((T, T), dtype=bool))
    hid = model.forward(tokens[:-1], list(range(T)), vis, model.new_cache())
    logits = model.logits(hid[n_prompt - 1:])
    worst = 0.0
    for j in range(logits.shape[0]):
        row = logits[j]
        best = row.max().item()
        deficit = best - row[tokens[n_prompt + j]].item()
        worst = max(worst, deficit - rel * abs(best))
        if deficit > tol + rel * abs(best):
            return False, (j, deficit, best)
    return True, worst


def _kv_rows(eng, n):
    return [eng.k_cache(li)[:, :n].float().clone() for li in range(eng.L)], [eng.vt_cache(li)[:, :, :n].float().clone() for li in range(eng.L)]


def _assert_cache_equals_plain_prefill(eng, tokens, n_rows, tag):
    """the K/V rows of the accepted prefix left behind by a lookahead run (appended step by step, moved by lade_kv_commit after
    every hit; lade/decoding.py:1145-1163) vs the rows ONE causal pass over the same tokens writes.  Same values up to the
    16-bit rounding of different kernel shapes; a misplaced or stale row differs by the spread of the values themselves."""
    k_la, v_la = _kv_rows(eng, n_rows)
    eng.reset()
    eng.prefill(tokens[:n_rows], [n_rows - 1])
    k_pf, v_pf = _kv_rows(eng, n_rows)
    for li in range(eng.L):
        for a, b, what in ((k_la[li], k_pf[li], "K"), (v_la[li], v_pf[li], "V")):
            spread = b.std().item()
            d = (a - b).abs()
            assert d.max().item() <= 0.15 * spread + 0.02 and d.mean().item() <= 0.01 * spread + 1e-3, (tag, li, what, d.max().item(), d.mean().item(), spread)


FULL_WIDTH = [("llama2-7b", 4, 15, 5, 15), ("codellama-13b", 3, 20, 7, 20), ("llama2-70b", 2, 15, 5, 15)]


@pytest.mark.parametrize("shape,layers,W,N,G", FULL_WIDTH)
def test_full_width_real_weights_bf16_cold(shape, layers, W, N, G):
    """Untied random weights at the BASELINE widths (hidden / heads / GQA / vocab at full size, a few layers), bf16: attention,
    GEMMs and glue all contribute to every logit.  The lookahead stream (eager and hipGraph) must be the plain greedy stream
    of the same engine - or differ only where both are within the logit margin of the fp32 oracle - and every emitted token
    must be within that margin (0.06 logit units on logits of spread ~1.3; the measured worst case is printed)."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    cfg = make_config(shape, layers=layers)
    w = random_weights_torch(cfg, seed=0, dtype=torch.bfloat16, device="cuda")
    w_cpu = {k: v.float().cpu() for k, v in w.items()}
    eng = StepEngine(cfg, w, dtype=torch.bfloat16, device="cuda", max_seq=1024, max_T=512)
    del w
    prompt = torch.randint(3, cfg["vocab"], (96,), generator=torch.Generator().manual_seed(123)).tolist()
    n_new = 24
    plain = eng.plain_greedy(prompt, len(prompt) + n_new)
    ok, worst_plain = _oracle_margin(cfg, w_cpu, torch.bfloat16, plain, len(prompt), tol=0.06)
    assert ok, ("plain", shape, worst_plain)
    for use_graph in (False, True):
        dec = LookaheadDecoder(eng, W, N, G, use_graph=use_graph)
        out = dec.greedy(prompt, len(prompt) + n_new, rng=random.Random(1))
        if out.tokens != plain:
            ok, worst = _oracle_margin(cfg, w_cpu, torch.bfloat16, out.tokens, len(prompt), tol=0.06)
            assert ok, (shape, use_graph, worst)
        _assert_cache_equals_plain_prefill(eng, dec.tokens, dec.P, (shape, use_graph))
    print(f"[{shape}] worst margin deficit of the plain stream: {worst_plain:.4f}")


@pytest.mark.parametrize("shape,layers,W,N,G", FULL_WIDTH)
def test_full_width_real_weights_bf16_with_accepted_ngrams(shape, layers, W, N, G):
    """The accept path with live attention / MLP at the BASELINE widths: tied embeddings of larger scale (std 1.0, the same
    order as a layer's contribution to the residual stream) make the model copy-biased, so its greedy stream becomes
    periodic and the pool hits (S > 2), while every projection still feeds the residual stream.  Lookahead == plain greedy on the same engine, every token within the oracle margin, and the cache
    after the run (rows committed out of candidate rows) equals a plain causal prefill of the same tokens."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    cfg = make_config(shape, layers=layers)
    w = random_weights_torch(cfg, seed=0, dtype=torch.bfloat16, device="cuda")
    w["embed"] = (w["embed"].float() * (1.0 / 0.02)).bfloat16()
    w["lm_head"] = w["embed"]
    w_cpu = {k: v.float().cpu() for k, v in w.items()}
    eng = StepEngine(cfg, w, dtype=torch.bfloat16, device="cuda", max_seq=1024, max_T=512)
    del w
    prompt = [(7 * i) % 50 + 3 for i in range(96)]
    n_new = 48
    plain = eng.plain_greedy(prompt, len(prompt) + n_new)
    assert _oracle_margin(cfg, w_cpu, torch.bfloat16, plain, len(prompt), tol=0.06, rel=0.01)[0]
    for use_graph in (False, True):
        dec = LookaheadDecoder(eng, W, N, G, pool_from_prompt=True, use_graph=use_graph)
        out = dec.greedy(prompt, len(prompt) + n_new, rng=random.Random(1), keep_trace=True)
        if out.tokens != plain:
            assert _oracle_margin(cfg, w_cpu, torch.bfloat16, out.tokens, len(prompt), tol=0.06, rel=0.01)[0], (shape, use_graph)
        assert out.steps * 2 < out.generated, (shape, use_graph, out.steps)                  # S > 2: n-grams are accepted
        assert max(t["max_hit"] for t in out.trace) == N - 2
        _assert_cache_equals_plain_prefill(eng, dec.tokens, dec.P, (shape, use_graph))
