"""Pins the CPU oracle (oracle/lade_oracle.py) to fixtures produced by the REFERENCE's own code
(oracle/make_golden.py ran /root/reference/lade unmodified).  CPU only."""
import json
import os
import random

import numpy as np
import pytest
import torch

import lade_oracle as O
from lookaheaddecoding_amd.weights import make_config, random_weights_numpy

from conftest import GOLDEN


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def tm_json(tm):
    return {str(k): [list(t) for t in v] for k, v in tm.items()}


def oracle_model(run):
    cfg = make_config(run["model"], max_pos=run.get("max_pos", 512), **({"rope_scaling": run["rope_scaling"]} if run.get("rope_scaling") else {}))
    w = random_weights_numpy(cfg, seed=run["model_seed"], std=run["std"])
    return O.OracleLlama(cfg, {k: torch.as_tensor(v) for k, v in w.items()})


def test_pool_kats():
    d = load("pool_kat.json")
    n_ops = 0
    for case in d["cases"]:
        tm = {}
        for op in case["ops"]:
            if op["op"] == "update":
                O.update_token_map(tm, op["lst"], op["past"], op["new"], case["LEVEL"], case["W"], case["G"])
            elif op["op"] == "prompt":
                O.fill_pool_with_prompt(op["prompts"], tm, case["LEVEL"], case["G"])
            else:
                O.append_new_generated_pool(op["tokens"], tm, case["LEVEL"], case["G"])
            assert tm_json(tm) == op["after"]
            n_ops += 1
    assert n_ops > 80
    for fw in d["filter_window"]:
        w = list(fw["window"])
        it = iter(fw["reset_seq"])
        O.filter_window(w, fw["eos"], lambda: next(it))
        assert w == fw["after"]


def test_mask_closed_form_matches_reference():
    d = load("mask_cases.json")
    assert len(d["cases"]) > 150
    for c in d["cases"]:
        T, P = c["T"], c["P"]
        n_input = c["level_offset"] + 1
        lay = O.StepLayout(ids=[0] * T, positions=[0] * T, n_input=n_input, level_sizes=c["level_sizes"], lguess=c["lguess"],
                           is_prefill=False, window=c["level_sizes"][-1])
        assert lay.level_offset == c["level_offset"]
        m = O.dense_mask(lay, P, c["gs"])
        rows = [format(int("".join("1" if b else "0" for b in r), 2), "x") for r in m.tolist()]
        assert rows == c["rows"], c
    lay = O.StepLayout(ids=[0] * 9, positions=[0] * 9, n_input=3, level_sizes=[6], lguess=0, is_prefill=True, window=6)
    m = O.dense_mask(lay, 0, 3)
    assert [format(int("".join("1" if b else "0" for b in r), 2), "x") for r in m.tolist()] == d["prefill_T9"]


def test_flash_row_order_equals_the_reference_numpy_expression():
    """oracle.flash_row_order vs the expression the reference uses to permute tokens for its flash kernel
    (lade/models/modeling_llama.py:1483: np.append(all_past[0], np.array(all_past[1:]).transpose().flatten()))."""
    for n_input, ls, lguess in ((1, [14, 15, 15, 15], 60), (1, [4, 5], 6), (3, [16, 17, 17], 0), (2, [6, 2, 2, 2], 8), (1, [9], 0)):
        base = n_input
        levels = []
        for n in ls:
            levels.append(list(range(base, base + n)))
            base += n
        if len(levels) > 1:
            mid = np.append(np.array(levels[0]), np.array(levels[1:]).transpose().flatten()).tolist()
        else:
            mid = levels[0]
        T = n_input + sum(ls) + lguess
        want = list(range(n_input)) + [int(x) for x in mid] + list(range(T - lguess, T))
        assert O.flash_row_order(n_input, ls, lguess) == want


def _check_trace(res, run, rank_traces=None):
    assert res.tokens == run["tokens"]
    assert res.steps == run["steps"]
    if "generated" in run:
        assert res.generated == run["generated"]
    ref_trace = run["trace"] if rank_traces is None else rank_traces[0]
    assert len(res.trace) == len(ref_trace)
    for i, (mine, ref) in enumerate(zip(res.trace, ref_trace)):
        assert mine.ids == ref["ids"], i
        assert mine.positions == ref["positions"], i
        assert mine.level_sizes == ref["level_sizes"], i
        assert mine.lguess == ref["lguess"], i
        assert mine.n_input == ref["n_input"], i
        assert mine.kvcache_len == ref["kvcache_len"], i
        assert mine.step_len == ref["step_len"], i
        assert mine.P == ref["P"], i
        assert mine.first_guess == ref["out_argmax"], i
        if rank_traces is not None:
            for r, rt in enumerate(rank_traces):
                assert mine.rank_ids[r] == rt[i]["ids"], (i, r)
                assert mine.rank_positions[r] == rt[i]["positions"], (i, r)


def test_greedy_e2e_matches_reference_traces():
    d = load("e2e_greedy.json")
    assert len(d["runs"]) >= 10
    wide = load("e2e_greedy_wide.json")["runs"]          # BASELINE config 4's W = 20, N = 7, G = 20 (steps of up to 240 tokens)
    assert len(wide) == 2 and max(len(st["ids"]) for r in wide for st in r["trace"]) == 240
    for run in d["runs"] + wide:
        model = oracle_model(run)
        res = O.lookahead_greedy(model, run["prompt"], run["W"], run["N"], run["G"], run["max_length"], random.Random(run["seed"]),
                                 eos_token_id=run["eos"], pool_from_prompt=bool(run["pool_from_prompt"]))
        _check_trace(res, run)
        if run["plain"] is not None:
            assert run["equals_plain_greedy"]
            assert O.plain_greedy(model, run["prompt"], run["max_length"]) == run["plain"]
            assert res.tokens == run["plain"][:len(res.tokens)]


def test_unlimited_guess_set_never_verifies_like_the_reference():
    """GUESS_SET_SIZE = -1: the reference's loops gate verification on GUESS_SET_SIZE > 0 (lade/decoding.py:402, :948)."""
    d = load("e2e_unlimited.json")
    for run in d["runs"]:
        model = oracle_model(run)
        res = O.lookahead_greedy(model, run["prompt"], run["W"], run["N"], run["G"], run["max_length"], random.Random(run["seed"]),
                                 eos_token_id=run["eos"], pool_from_prompt=bool(run["pool_from_prompt"]))
        _check_trace(res, run)
        assert res.steps == res.generated


def test_dynamic_ntk_rope_matches_reference_traces():
    """rope_scaling = dynamic (LlamaDynamicNTKScalingRotaryEmbedding, lade/models/modeling_llama.py:292-318): max_position_embeddings far
    below the generated length, so the reference rebuilds its tables - new base - at nearly every step, on kv_seq_len = P + T with the
    longest length kept (one run's step lengths go 42, 39, 39, 40: no rebuild there).  Tokens, steps and every step's inputs."""
    d = load("e2e_dynamic_ntk.json")
    assert len(d["runs"]) == 2
    differs = []
    for run in d["runs"]:
        assert max(st["step_len"] for st in run["trace"]) > run["max_pos"] * 2
        model = oracle_model(run)
        res = O.lookahead_greedy(model, run["prompt"], run["W"], run["N"], run["G"], run["max_length"], random.Random(run["seed"]),
                                 pool_from_prompt=bool(run["pool_from_prompt"]))
        _check_trace(res, run)
        # and the tables matter: the same run with the scaling switched off parts from the reference
        plain_rope = O.OracleLlama(make_config(run["model"], max_pos=512), model.w)
        other = O.lookahead_greedy(plain_rope, run["prompt"], run["W"], run["N"], run["G"], run["max_length"], random.Random(run["seed"]),
                                   pool_from_prompt=bool(run["pool_from_prompt"]))
        differs.append(other.tokens != run["tokens"])
    assert any(differs)            # (a run whose tiny model is insensitive to the rotation may coincide; not both)


def test_dynamic_ntk_state_persists_across_calls_like_the_reference_module():
    """Two consecutive generate() calls on ONE reference model (tests/golden/e2e_dynamic_ntk_again.json): `max_seq_len_cached` and the rebuilt
    inv_freq live on the rotary module and are never reset (lade/models/modeling_llama.py:243-246, :299-316), so the second call rotates with
    the largest base the first one reached - and emits OTHER tokens than a fresh model would.  The oracle's model object keeps the state
    the same way."""
    d = load("e2e_dynamic_ntk_again.json")
    assert len(d["runs"]) == 2
    for run in d["runs"]:
        model = oracle_model(run)
        for call in run["calls"]:
            res = O.lookahead_greedy(model, call["prompt"], run["W"], run["N"], run["G"], call["max_length"], random.Random(run["seed"]))
            assert res.tokens == call["tokens"] and res.steps == call["steps"], run["model"]
        c2 = run["calls"][1]
        fresh = O.lookahead_greedy(oracle_model(run), c2["prompt"], run["W"], run["N"], run["G"], c2["max_length"], random.Random(run["seed"]))
        assert fresh.tokens == run["second_call_on_a_fresh_model"]["tokens"] and fresh.tokens != c2["tokens"]


def test_lookahead_parallel_matches_reference_gloo_runs():
    d = load("e2e_lp.json")
    for run in d["runs"]:
        model = oracle_model(run)
        res = O.lookahead_greedy(model, run["prompt"], run["W"], run["N"], run["G"], run["max_length"], random.Random(run["seed"]),
                                 R=run["R"], pool_from_prompt=bool(run.get("pool_from_prompt", 0)), eos_token_id=run.get("eos"))
        _check_trace(res, run, run["rank_traces"])


def test_sampling_e2e_matches_reference():
    d = load("e2e_sample.json")
    for run in d["runs"]:
        model = oracle_model(run)
        torch.manual_seed(run["seed"])
        res = O.lookahead_sample(model, run["prompt"], run["W"], run["N"], run["G"], run["max_length"], random.Random(run["seed"]),
                                 torch.default_generator, **run["warp"])
        assert res.tokens == run["tokens"], run["warp"]
        assert res.steps == run["steps"]
        for mine, ref in zip(res.trace, run["trace"]):
            assert mine.ids == ref["ids"]
            assert mine.positions == ref["positions"]


def test_sampling_with_eos_and_pool_from_prompt_matches_reference():
    """filter_window on the newest level, the EOS stop and POOL_FROM_PROMPT inside the sampling loop (lade/decoding.py:578-604)."""
    d = load("e2e_sample_eos.json")
    for run in d["runs"]:
        model = oracle_model(run)
        torch.manual_seed(run["seed"])
        res = O.lookahead_sample(model, run["prompt"], run["W"], run["N"], run["G"], run["max_length"], random.Random(run["seed"]),
                                 torch.default_generator, eos_token_id=run["eos"], pool_from_prompt=bool(run["pool_from_prompt"]), **run["warp"])
        assert res.tokens == run["tokens"], (run["warp"], run["eos"])
        assert res.steps == run["steps"]
        for mine, ref in zip(res.trace, run["trace"]):
            assert mine.ids == ref["ids"] and mine.positions == ref["positions"]


def test_attention_layer_matches_reference_capture():
    """RoPE + dense attention of the oracle against tensors captured inside the reference's own
    LlamaAttention.forward (q/k/v projections, post-step K/V cache, o_proj input)."""
    z = np.load(os.path.join(GOLDEN, "attn_steps.npz"))
    d = load("e2e_greedy.json")
    keys = sorted({k.rsplit(".", 2)[0] for k in z.files})
    assert len(keys) >= 4
    for base in keys:   # e.g. tiny-d64.5.3.3.s1.step5
        mname, W, N, G, seed, step = base.split(".")
        W, N, G, seed, step = int(W), int(N), int(G), int(seed[1:]), int(step[4:])
        run = [r for r in d["runs"] if r["model"] == mname and (r["W"], r["N"], r["G"], r["seed"]) == (W, N, G, seed) and r["eos"] is None][0]
        tr = run["trace"][step]
        cfg = make_config(mname, max_pos=512)
        H, Hkv, dh = cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
        cos, sin = O.rope_tables(dh, 512, cfg["rope_theta"])
        lay = O.StepLayout(ids=tr["ids"], positions=tr["positions"], n_input=tr["n_input"], level_sizes=tr["level_sizes"],
                           lguess=tr["lguess"], is_prefill=tr["is_prefill"], window=0)
        vis = O.dense_mask(lay, tr["P"], N - 1)
        T = lay.T
        pos = torch.as_tensor(tr["positions"])
        for li in range(cfg["layers"]):
            q = torch.as_tensor(z[f"{base}.L{li}.q_proj"]).view(T, H, dh).transpose(0, 1)
            k = torch.as_tensor(z[f"{base}.L{li}.k_proj"]).view(T, Hkv, dh).transpose(0, 1)
            K = torch.as_tensor(z[f"{base}.L{li}.K"])
            V = torch.as_tensor(z[f"{base}.L{li}.V"])
            assert K.shape[1] == tr["P"] + T
            qr = O.apply_rope(q, cos, sin, pos)
            kr = O.apply_rope(k, cos, sin, pos)
            assert torch.allclose(kr, K[:, tr["P"]:], atol=1e-6)
            o = O.attention_dense(qr, K, V, vis).transpose(0, 1).reshape(T, H * dh)
            ref = torch.as_tensor(z[f"{base}.L{li}.attn_out"])
            assert torch.allclose(o, ref, atol=2e-6, rtol=1e-5), (base, li, (o - ref).abs().max())
