"""End-to-end parity of the HIP lookahead loop: identical greedy token ids AND identical step
counts / per-step acceptance against the reference-generated traces (fp32), plus the
lookahead == plain-greedy property and an oracle scoring check at bf16."""
import json
import os
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

import lade_oracle as O
from conftest import GOLDEN
from lookaheaddecoding_amd.weights import make_config, random_weights_numpy


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def make_engine(run_or_name, dtype, seed=None, std=None, max_seq=512):
    from lookaheaddecoding_amd.engine import StepEngine
    if isinstance(run_or_name, dict):
        name, seed, std = run_or_name["model"], run_or_name["model_seed"], run_or_name["std"]
    else:
        name = run_or_name
    cfg = make_config(name, max_pos=512)
    w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=seed, std=std).items()}
    return cfg, w, StepEngine(cfg, w, dtype=dtype, max_seq=max_seq, max_T=320)


def test_greedy_fp32_identical_tokens_steps_and_trace_vs_reference():
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    d = load("e2e_greedy.json")
    n = 0
    for run in d["runs"] + load("e2e_greedy_wide.json")["runs"]:          # + config 4's W = 20, N = 7, G = 20 (steps of up to 240 tokens)
        cfg, w, eng = make_engine(run, torch.float32)
        dec = LookaheadDecoder(eng, run["W"], run["N"], run["G"], pool_from_prompt=bool(run["pool_from_prompt"]))
        out = dec.greedy(run["prompt"], run["max_length"], eos_token_id=run["eos"], rng=random.Random(run["seed"]), keep_trace=True)
        assert out.tokens == run["tokens"], (run["model"], run["W"], run["N"], run["G"], run["seed"])
        assert out.steps == run["steps"] and out.generated == run["generated"]
        for i, (mine, ref) in enumerate(zip(out.trace, run["trace"])):
            assert mine["T"] == len(ref["ids"]) and mine["P_before"] == ref["P"] and mine["first_guess"] == ref["out_argmax"], i
            if i + 1 < len(run["trace"]):
                assert mine["P_after"] == run["trace"][i + 1]["P"], i
        if run["plain"] is not None:
            assert eng.plain_greedy(run["prompt"], run["max_length"]) == run["plain"]
        n += 1
    assert n >= 10


def test_unlimited_guess_set_matches_reference():
    """GUESS_SET_SIZE = -1 ("unlimited"): the reference never enters the verification branch; eager and graph mode."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    for run in load("e2e_unlimited.json")["runs"]:
        cfg, w, eng = make_engine(run, torch.float32)
        for use_graph in (False, True):
            dec = LookaheadDecoder(eng, run["W"], run["N"], run["G"], pool_from_prompt=bool(run["pool_from_prompt"]), use_graph=use_graph)
            out = dec.greedy(run["prompt"], run["max_length"], rng=random.Random(run["seed"]), keep_trace=True)
            assert out.tokens == run["tokens"] and out.steps == run["steps"] == out.generated
            for i, (mine, ref) in enumerate(zip(out.trace, run["trace"])):
                assert mine["T"] == len(ref["ids"]) and mine["P_before"] == ref["P"] and mine["first_guess"] == ref["out_argmax"], (i, use_graph)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_greedy_16bit_equals_plain_greedy_and_scores_within_tolerance(dtype):
    """Output-identity property of lookahead decoding (README.md:132) on the MFMA path, and every emitted
    token is within tolerance of the fp32 oracle's best logit for its prefix."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    for (name, seed, std, W, N, G, prompt) in (("tiny-d64", 1, 0.05, 5, 3, 3, [1, 5, 9, 17, 33, 5, 9, 17, 44, 5, 9]),
                                               ("tiny-d128", 2, 0.05, 15, 5, 15, [1, 5, 9, 17, 33, 5, 9, 17, 44, 5, 9]),
                                               ("tiny-d64", 1, 0.05, 7, 5, 7, [3, 7, 7, 3, 7, 7, 3, 11, 7, 7, 3, 7])):
        cfg, w, eng = make_engine(name, dtype, seed, std)
        dec = LookaheadDecoder(eng, W, N, G)
        max_length = len(prompt) + 48
        out = dec.greedy(prompt, max_length, rng=random.Random(1), keep_trace=True)
        plain = eng.plain_greedy(prompt, max_length)
        margin_ok = _oracle_margin_check(cfg, w, dtype, out.tokens, len(prompt), tol=0.06 if dtype == torch.bfloat16 else 0.02)
        assert margin_ok
        if out.tokens != plain:   # a 16-bit near-tie may flip; then both streams must still be oracle-valid
            assert _oracle_margin_check(cfg, w, dtype, plain, len(prompt), tol=0.06 if dtype == torch.bfloat16 else 0.02)
        assert out.steps < out.generated, "no n-gram was ever accepted: the hot regime was not exercised"


def _oracle_margin_check(cfg, w, dtype, tokens, n_prompt, tol):
    wq = {k: v.to(dtype).float() for k, v in w.items()}
    model = O.OracleLlama(cfg, wq)
    import numpy as np
    T = len(tokens) - 1
    vis = np.tril(np.ones((T, T), dtype=bool))
    hid = model.forward(tokens[:-1], list(range(T)), vis, model.new_cache())
    logits = model.logits(hid)
    for i in range(n_prompt - 1, T):
        row = logits[i]
        if row.max().item() - row[tokens[i + 1]].item() > tol:
            return False
    return True


def test_graph_mode_identical_to_reference_traces_fp32():
    """steady steps replayed as one captured hipGraph (fixed T_max, padded candidates, P read on device):
    same tokens, same step count, same per-step acceptance as the reference."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    d = load("e2e_greedy.json")
    for run in d["runs"] + load("e2e_greedy_wide.json")["runs"]:
        cfg, w, eng = make_engine(run, torch.float32)
        dec = LookaheadDecoder(eng, run["W"], run["N"], run["G"], pool_from_prompt=bool(run["pool_from_prompt"]), use_graph=True)
        out = dec.greedy(run["prompt"], run["max_length"], eos_token_id=run["eos"], rng=random.Random(run["seed"]), keep_trace=True)
        assert out.tokens == run["tokens"], (run["model"], run["W"], run["N"], run["G"], run["seed"])
        assert out.steps == run["steps"]
        for i, (mine, ref) in enumerate(zip(out.trace, run["trace"])):
            assert mine["P_before"] == ref["P"] and mine["first_guess"] == ref["out_argmax"], i


def test_graph_mode_bf16_equals_eager_mode():
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    prompt = [1, 5, 9, 17, 33, 5, 9, 17, 44, 5, 9]
    cfg, w, eng = make_engine("tiny-d128", torch.bfloat16, 2, 0.05)
    a = LookaheadDecoder(eng, 15, 5, 15).greedy(prompt, len(prompt) + 60, rng=random.Random(1), keep_trace=True)
    b = LookaheadDecoder(eng, 15, 5, 15, use_graph=True).greedy(prompt, len(prompt) + 60, rng=random.Random(1), keep_trace=True)
    assert a.tokens == b.tokens and a.steps == b.steps
    assert [t["max_hit"] for t in a.trace] == [t["max_hit"] for t in b.trace]


def test_long_generation_equals_plain_greedy_and_the_oracle_fp32():
    """1200 new tokens (graph and eager mode): thousands of pool inserts with evictions, every candidate bucket, a KV
    cache that grows past several attention-split regimes - the stream must stay the plain greedy stream, and the first
    200 tokens and their step count must be the CPU oracle's (the oracle is pinned to the reference)."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    cfg, w, eng = make_engine("tiny-d64", torch.float32, 1, 0.05, max_seq=2048)
    prompt = [1, 5, 9, 17, 33, 5, 9, 17, 44, 5, 9, 2, 5, 9]
    n_new = 1200
    plain = eng.plain_greedy(prompt, len(prompt) + n_new)
    outs = []
    for use_graph in (False, True):
        dec = LookaheadDecoder(eng, 7, 5, 7, pool_from_prompt=True, use_graph=use_graph)
        out = dec.greedy(prompt, len(prompt) + n_new, rng=random.Random(3), keep_trace=True)
        assert out.tokens == plain, use_graph
        assert out.steps < n_new            # the pool does hit on this repetitive stream
        outs.append(out)
    assert outs[0].steps == outs[1].steps and [t["max_hit"] for t in outs[0].trace] == [t["max_hit"] for t in outs[1].trace]
    model = O.OracleLlama(cfg, {k: torch.as_tensor(v) for k, v in w.items()})
    ref = O.lookahead_greedy(model, prompt, 7, 5, 7, len(prompt) + 200, random.Random(3), pool_from_prompt=True, keep_trace=True)
    assert ref.tokens == plain[:len(ref.tokens)]
    n_ref = ref.steps
    assert [t["max_hit"] for t in outs[0].trace[:n_ref - 1]] == [t.max_hit for t in ref.trace[:n_ref - 1]]


def test_reference_default_configuration_w60_n8_g60():
    """config_lade never called: the reference falls back to WINDOW_SIZE=60, LEVEL=8, GUESS_SET_SIZE=60
    (lade/decoding.py:854-857) - steps of up to 847 tokens.  Tokens, steps and acceptance pattern vs the oracle (fp32), and
    the bf16 MFMA path against plain greedy on the same kernels."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    prompt = [1, 5, 9, 17, 33, 5, 9, 17, 44, 5, 9, 2, 5, 9, 17]
    cfg, w, eng = make_engine("tiny-d64", torch.float32, 1, 0.05, max_seq=2048)
    eng = type(eng)(cfg, w, dtype=torch.float32, max_seq=2048, max_T=896)
    model = O.OracleLlama(cfg, {k: torch.as_tensor(v) for k, v in w.items()})
    ref = O.lookahead_greedy(model, prompt, 60, 8, 60, len(prompt) + 48, random.Random(2), keep_trace=True)
    for use_graph in (False, True):
        out = LookaheadDecoder(eng, 60, 8, 60, use_graph=use_graph).greedy(prompt, len(prompt) + 48, rng=random.Random(2), keep_trace=True)
        assert out.tokens == ref.tokens and out.steps == ref.steps, use_graph
        assert [t["max_hit"] for t in out.trace] == [t.max_hit for t in ref.trace]
    cfg, w, eng16 = make_engine("tiny-d128", torch.bfloat16, 2, 0.05, max_seq=2048)
    eng16 = type(eng16)(cfg, w, dtype=torch.bfloat16, max_seq=2048, max_T=896)
    out = LookaheadDecoder(eng16, 60, 8, 60).greedy(prompt, len(prompt) + 40, rng=random.Random(2))
    assert out.tokens == eng16.plain_greedy(prompt, len(prompt) + 40)


def test_graph_recapture_when_the_cache_outgrows_its_split_count():
    """bf16, 2600 new tokens in graph mode: the KV split count of the captured attention follows the cache length
    (re-capture), the run completes, and its head is the eager run's - or parts from it only at a rounding near-tie."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    cfg, w, eng = make_engine("tiny-d128", torch.bfloat16, 2, 0.05, max_seq=4096)
    prompt = [1, 5, 9, 17, 33, 5, 9, 17, 44, 5, 9]
    dec = LookaheadDecoder(eng, 7, 4, 7, use_graph=True)
    captures = []
    orig = dec._capture_graphs
    dec._capture_graphs = lambda *a, **k: (captures.append(dec.P), orig(*a, **k))[1]
    out = dec.greedy(prompt, len(prompt) + 2600, rng=random.Random(5))
    assert out.generated == 2600 and all(0 <= t < cfg["vocab"] for t in out.tokens)
    assert len(captures) >= 2 and captures[-1] > 1024, captures
    eager = LookaheadDecoder(eng, 7, 4, 7).greedy(prompt, len(prompt) + 300, rng=random.Random(5))
    # Same kernels, same split COUNT - but the graph step pads its candidate slots, so the key count P + T and with it the boundaries of
    # the contiguous KV splits can differ from the eager step's by one tile: 16-bit partials then round differently, and the two greedy
    # streams may part at a near-tie.  Where they do, both must still be greedy streams of the model (fp32 oracle margin).
    head = out.tokens[:len(eager.tokens)]
    if eager.tokens != head:
        assert _oracle_margin_check(cfg, w, torch.bfloat16, eager.tokens, len(prompt), tol=0.06)
        assert _oracle_margin_check(cfg, w, torch.bfloat16, head, len(prompt), tol=0.06)


def test_sampling_fp32_identical_tokens_vs_reference():
    """jacobi_sample_multilevel parity: same python/torch RNG order, probabilities from the HIP step in fp32 ->
    the reference's sampled token ids and step counts (temperature / top-k / top-p runs)."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.sampling import make_warper
    d = load("e2e_sample.json")
    for run in d["runs"]:
        cfg, w, eng = make_engine(run, torch.float32)
        for use_graph in (False, True):             # True: steady steps replay the forward-only hipGraph
            dec = LookaheadDecoder(eng, run["W"], run["N"], run["G"], use_graph=use_graph)
            torch.manual_seed(run["seed"])
            out = dec.sample(run["prompt"], run["max_length"], warp=make_warper(**run["warp"]), rng=random.Random(run["seed"]),
                             torch_gen=torch.default_generator)
            assert out.tokens == run["tokens"], (run["warp"], use_graph)
            assert out.steps == run["steps"]


def test_sampling_with_eos_and_pool_from_prompt_vs_reference():
    """EOS replacement in the newest window level (filter_window), the EOS stop and POOL_FROM_PROMPT in the sampling loop."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.sampling import make_warper
    for run in load("e2e_sample_eos.json")["runs"]:
        cfg, w, eng = make_engine(run, torch.float32)
        for use_graph in (False, True):
            dec = LookaheadDecoder(eng, run["W"], run["N"], run["G"], pool_from_prompt=bool(run["pool_from_prompt"]), use_graph=use_graph)
            torch.manual_seed(run["seed"])
            out = dec.sample(run["prompt"], run["max_length"], warp=make_warper(**run["warp"]), eos_token_id=run["eos"],
                             rng=random.Random(run["seed"]), torch_gen=torch.default_generator)
            assert out.tokens == run["tokens"], (run["warp"], run["eos"], use_graph)
            assert out.steps == run["steps"]


def test_sampling_logits_within_tolerance_bf16():
    """north_star: 'sampling logits match within a stated fp tolerance' - bf16 step logits vs the fp32 oracle on
    the same (bf16-rounded) weights, steady step with candidates: atol 6e-2 on logits of magnitude ~1."""
    from lookaheaddecoding_amd.ops import StepMask
    cfg, w, eng = make_engine("tiny-d128", torch.bfloat16, 2, 0.05)
    wq = {k: v.bfloat16().float() for k, v in w.items()}
    model = O.OracleLlama(cfg, wq)
    torch.manual_seed(0)
    prompt = torch.randint(3, cfg["vocab"], (40,)).tolist()
    W, N, gs, g = 15, 5, 4, 6
    past = [[int(x) for x in torch.randint(3, cfg["vocab"], (W - 1,))]] + [[int(x) for x in torch.randint(3, cfg["vocab"], (W,))] for _ in range(N - 2)]
    guess = [int(x) for x in torch.randint(3, cfg["vocab"], (g * gs,))]
    cache = model.new_cache()
    import numpy as np
    P = len(prompt)
    vis = np.tril(np.ones((P, P), dtype=bool))
    model.forward(prompt, list(range(P)), vis, cache)
    ref = O.model_step(model, cache, [7], [P], past, guess, N - 2, gs)
    # engine: prefill then the same step
    ids = torch.tensor(prompt, dtype=torch.int32, device="cuda"); pos = torch.arange(P, dtype=torch.int32, device="cuda")
    eng.forward(ids, pos, StepMask(T=P, P=0, is_prefill=True), torch.zeros(1, dtype=torch.int32, device="cuda"), 0)
    lay = ref.layout
    T = lay.T
    sel = torch.arange(T, dtype=torch.int32, device="cuda")
    logits = eng.forward(torch.tensor(lay.ids, dtype=torch.int32, device="cuda"), torch.tensor(lay.positions, dtype=torch.int32, device="cuda"),
                         StepMask.from_levels(1, lay.level_sizes, lay.lguess, gs, P), sel, T).float().cpu()
    assert torch.allclose(logits[0], ref.out_logits, atol=6e-2, rtol=5e-2)
    assert torch.allclose(logits[T - lay.lguess - W:T - lay.lguess], ref.inp_logits, atol=6e-2, rtol=5e-2)
    assert torch.allclose(logits[T - lay.lguess:], ref.guess_logits, atol=6e-2, rtol=5e-2)


@pytest.mark.parametrize("use_graph", [False, True])
def test_cache_exhaustion_raises_and_never_writes_past_the_cache(use_graph):
    """A tightly sized KV cache: the run must end with LadeHipError when a step no longer fits - in hipGraph mode too, where the
    capture warm-up runs whole step bodies at the current cache length and the kernels read the length from the device - and
    everything decoded before that must still be the plain greedy stream (no K/V row was written over a neighbouring head)."""
    from lookaheaddecoding_amd import cabi
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    cfg = make_config("tiny-d128", max_pos=512)
    w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=2, std=0.05).items()}
    eng = StepEngine(cfg, w, dtype=torch.float32, max_seq=128, max_T=64)
    assert eng.S_max == 128
    prompt = [1, 5, 9, 17, 33, 5, 9, 17, 44, 5, 9] * 3
    dec = LookaheadDecoder(eng, 5, 4, 5, pool_from_prompt=True, use_graph=use_graph)
    with pytest.raises(cabi.LadeHipError, match="exhausted|exceeds"):
        dec.greedy(prompt, 400, rng=random.Random(2))
    got = list(dec.tokens)
    assert len(got) > len(prompt) + 40                      # it ran until the cache was nearly full
    big = StepEngine(cfg, w, dtype=torch.float32, max_seq=512, max_T=64)
    assert big.plain_greedy(prompt, len(got)) == got


def test_dynamic_ntk_rope_matches_reference_traces_eager_and_graph():
    """rope_scaling = dynamic (lade/models/modeling_llama.py:292-318) on the HIP step: `lade_rope_rows_dynamic` keeps the reference's
    "longest kv_seq_len seen" on the device and writes the step's cos / sin rows from it - fp32 engine vs the reference's own runs (tables
    rebuilt at nearly every step, one run with non-monotone step lengths), eager and hipGraph (whose padded candidate slots must not count
    as sequence length), tokens / steps / per-step cache lengths.  Each run starts from a fresh model's state (reset_rope_state): the
    state itself survives reset(), as the reference's does across generate() calls (next test)."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    for run in load("e2e_dynamic_ntk.json")["runs"]:
        cfg = make_config(run["model"], max_pos=run["max_pos"], rope_scaling=run["rope_scaling"])
        w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=run["model_seed"], std=run["std"]).items()}
        eng = StepEngine(cfg, w, dtype=torch.float32, max_seq=512, max_T=320)
        assert eng.ntk_state is not None and int(eng.ntk_state.item()) == run["max_pos"]
        for use_graph in (False, True, False):
            eng.reset_rope_state()
            dec = LookaheadDecoder(eng, run["W"], run["N"], run["G"], pool_from_prompt=bool(run["pool_from_prompt"]), use_graph=use_graph)
            out = dec.greedy(run["prompt"], run["max_length"], rng=random.Random(run["seed"]), keep_trace=True)
            assert out.tokens == run["tokens"] and out.steps == run["steps"], (run["model"], use_graph)
            for i, (mine, ref) in enumerate(zip(out.trace, run["trace"])):
                assert mine["T"] >= len(ref["ids"]) and mine["P_before"] == ref["P"] and mine["first_guess"] == ref["out_argmax"], (i, use_graph)
            assert int(eng.ntk_state.item()) == max(st["step_len"] for st in run["trace"])


def test_dynamic_ntk_state_survives_reset_like_the_reference_module():
    """Two consecutive generate() calls on ONE reference model (tests/golden/e2e_dynamic_ntk_again.json): the rotary module's longest length
    seen is never reset (lade/models/modeling_llama.py:243-246, :299-316), the second call rotates with the first call's largest base and
    emits other tokens than a fresh model.  One engine, two greedy() calls == the reference's two calls (eager and hipGraph); after
    reset_rope_state() the second prompt yields the fresh model's tokens."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    for run in load("e2e_dynamic_ntk_again.json")["runs"]:
        cfg = make_config(run["model"], max_pos=run["max_pos"], rope_scaling=run["rope_scaling"])
        w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=run["model_seed"], std=run["std"]).items()}
        for use_graph in (False, True):
            eng = StepEngine(cfg, w, dtype=torch.float32, max_seq=512, max_T=320)
            for call in run["calls"]:
                dec = LookaheadDecoder(eng, run["W"], run["N"], run["G"], use_graph=use_graph)
                out = dec.greedy(call["prompt"], call["max_length"], rng=random.Random(run["seed"]))
                assert out.tokens == call["tokens"] and out.steps == call["steps"], (run["model"], use_graph)
            assert int(eng.ntk_state.item()) == max(c["longest_step"] for c in run["calls"])
            eng.reset_rope_state()
            c2 = run["calls"][1]
            out = LookaheadDecoder(eng, run["W"], run["N"], run["G"], use_graph=use_graph).greedy(c2["prompt"], c2["max_length"], rng=random.Random(run["seed"]))
            assert out.tokens == run["second_call_on_a_fresh_model"]["tokens"] != c2["tokens"]
