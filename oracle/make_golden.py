"""Generates tests/golden/* by running the REFERENCE's own code (unmodified, /root/reference) on CPU.

Run in the build container only (`python oracle/make_golden.py`); the fixtures it writes are
committed, because /root/reference does not exist on the GPU box.  TEST INFRASTRUCTURE.

Fixtures (all produced by reference functions, nothing by our code, except the synthetic
weights which are inputs):
  pool_kat.json      update_token_map / fill_pool_with_prompt / append_new_generated_pool /
                     filter_window                         (lade/decoding.py:37-135)
  mask_cases.json    j_make_causal_mask_multilevel          (lade/models/modeling_llama.py:115-207)
  e2e_greedy.json    jacobi_greedy_search_multilevel traces (lade/decoding.py:697-1259)
  e2e_lp.json        the same under lookahead parallelism, gloo, R=2,3
  e2e_sample.json    jacobi_sample_multilevel traces        (lade/decoding.py:137-692)
  attn_steps.npz     per-layer q/k/v projections, K/V cache and attention output of real steps
                     (lade/models/modeling_llama.py:461-563)
"""
import json
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(1, ROOT)

import numpy as np
import torch

import ref_shim

lade = ref_shim.load_reference()
import lade.decoding as D
import lade.models.modeling_llama as M
from transformers import GenerationConfig, GenerationMixin, LlamaConfig
from transformers.generation.logits_process import (LogitsProcessorList, TemperatureLogitsWarper, TopKLogitsWarper,
                                                    TopPLogitsWarper)
from transformers.generation.stopping_criteria import MaxLengthCriteria, StoppingCriteriaList

from lookaheaddecoding_amd.weights import make_config, random_weights_numpy

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)


def dump(name, obj):
    with open(os.path.join(GOLD, name), "w") as f:
        json.dump(obj, f, separators=(",", ":"))
    print("wrote", name, os.path.getsize(os.path.join(GOLD, name)), "bytes")


def tm_to_json(tm):
    return {str(k): [list(t) for t in v] for k, v in tm.items()}

# ------------------------------------------------------------------ pool KATs


def gen_pool():
    rs = random.Random(1234)
    cases = []
    for (LEVEL, W, G, vocab) in [(4, 3, 2, 40), (5, 15, 15, 16), (3, 5, 3, 8), (7, 20, 20, 12), (5, 7, 7, 1000), (4, 5, 1, 6)]:
        tm = {}
        ops = []
        for it in range(14):
            kind = rs.choice(["update"] * 5 + ["prompt", "append"])
            if kind == "update":
                lst = rs.randrange(vocab)
                past = [[rs.randrange(vocab) for _ in range(W - 1)]] + [[rs.randrange(vocab) for _ in range(W)] for _ in range(LEVEL - 2)]
                new = [rs.randrange(vocab) for _ in range(W)]
                D.update_token_map(tm, lst, past, new, LEVEL, W, G)
                ops.append({"op": "update", "lst": lst, "past": past, "new": new, "after": tm_to_json(tm)})
            elif kind == "prompt":
                pr = [rs.randrange(vocab) for _ in range(rs.randrange(0, 30))]
                D.fill_pool_with_prompt(pr, tm, LEVEL, G)
                ops.append({"op": "prompt", "prompts": pr, "after": tm_to_json(tm)})
            else:
                toks = [rs.randrange(vocab) for _ in range(rs.choice([LEVEL, LEVEL, LEVEL - 1]))]
                D.append_new_generated_pool(toks, tm, LEVEL, G)
                ops.append({"op": "append", "tokens": toks, "after": tm_to_json(tm)})
        cases.append({"LEVEL": LEVEL, "W": W, "G": G, "ops": ops})
    # the SURVEY Appendix-A KATs, re-executed
    tm = {}
    kat = []
    for lst, past, new in [(7, [[10, 11], [20, 21, 22], [30, 31, 32]], [40, 41, 42]),
                           (7, [[10, 11], [50, 51, 52], [60, 61, 62]], [70, 71, 72]),
                           (7, [[10, 11], [80, 81, 82], [90, 91, 92]], [100, 101, 102]),
                           (7, [[10, 11], [50, 51, 52], [60, 61, 62]], [70, 71, 72])]:
        D.update_token_map(tm, lst, past, new, 4, 3, 2)
        kat.append({"op": "update", "lst": lst, "past": past, "new": new, "after": tm_to_json(tm)})
    cases.append({"LEVEL": 4, "W": 3, "G": 2, "ops": kat})
    tm = {}
    D.update_token_map(tm, 10, [[10, 10], [1, 2, 3], [4, 5, 6]], [7, 8, 9], 4, 3, 2)
    cases.append({"LEVEL": 4, "W": 3, "G": 2, "ops": [{"op": "update", "lst": 10, "past": [[10, 10], [1, 2, 3], [4, 5, 6]],
                                                      "new": [7, 8, 9], "after": tm_to_json(tm)}]})
    # filter_window
    fw = []
    for _ in range(6):
        lw = [rs.randrange(5) for _ in range(rs.randrange(1, 12))]
        seq = [100 + i for i in range(20)]
        it = iter(seq)
        after = list(lw)
        D.filter_window(after, 2, lambda: next(it))
        fw.append({"window": lw, "eos": 2, "reset_seq": seq, "after": after})
    dump("pool_kat.json", {"cases": cases, "filter_window": fw})

# ------------------------------------------------------------------ masks


def gen_masks():
    cases = []
    fmin = torch.finfo(torch.float32).min

    def one(level_sizes, lguess, gs, level_offset, P):
        T = level_offset + 1 + sum(level_sizes) + lguess
        guess = list(range(lguess)) if lguess > 0 else None
        m = M.j_make_causal_mask_multilevel(level_sizes, False, 0, guess, gs, 0, 0, (1, T), torch.float32, 0,
                                            torch.device("cpu"), past_key_values_length=P)
        m = m[0, 0]
        assert m.shape == (T, P + T)
        vis = (m == 0)
        assert ((m == 0) | (m == fmin)).all()
        rows = [format(int("".join("1" if b else "0" for b in row.tolist()), 2), "x") for row in vis]
        cases.append({"level_sizes": level_sizes, "lguess": lguess, "gs": gs, "level_offset": level_offset, "P": P,
                      "T": T, "rows": rows})

    for N in (2, 3, 4, 5, 7, 9):
        gs = N - 1
        for W in (1, 2, 5, 7):
            steady = [W - 1] + [W] * (N - 2)
            for g in (0, 1, 3):
                for lo in (0, 2):
                    if N == 2 and W == 1:
                        continue
                    one(steady, g * gs, gs, lo, 3)
            # fill phase shapes: level sizes after k fill steps (lade/decoding.py:1040-1062)
            for k in range(1, N - 1):
                l0 = W + N - 3 - k
                one([l0] + [l0 + 1] * k, 0, gs, 0, 2)
            # lookahead-parallel shards (dist_offset = window_start > 0)
            for (R, r) in ((2, 1), (3, 1), (3, 2)):
                wl = W
                split = (wl + R - 1) // R
                ws, we = min(split * r, wl), min(split * (r + 1), wl)
                if we - ws <= 0 or N < 3:
                    continue
                one([we - 1] + [we - ws] * (N - 2), 2 * gs, gs, 0, 1)
                one([we - 1] + [we - ws] * (N - 2), 0, gs, 3, 1)
    one([14, 15, 15, 15], 60, 4, 0, 5)        # BASELINE config 2 steady state, g = G
    one([4, 5, 5], 6, 3, 0, 3)                # README figure (N=4 W=5, two candidates)
    one([5, 5, 5], 6, 3, 0, 3)                # SURVEY Appendix A rendering
    # prefill = plain causal
    m = M.j_make_causal_mask_multilevel([6], True, 0, None, 3, 0, 0, (1, 9), torch.float32, 0, torch.device("cpu"), 0)[0, 0]
    dump("mask_cases.json", {"cases": cases, "prefill_T9": [format(int("".join("1" if b else "0" for b in r.tolist()), 2), "x") for r in (m == 0)]})

# ------------------------------------------------------------------ reference model on synthetic weights


class RefLM(M.LlamaForCausalLM, GenerationMixin):
    pass


def build_ref_model(cfg, weights):
    hc = LlamaConfig(vocab_size=cfg["vocab"], hidden_size=cfg["hidden"], intermediate_size=cfg["inter"],
                     num_hidden_layers=cfg["layers"], num_attention_heads=cfg["heads"], num_key_value_heads=cfg["kv_heads"],
                     max_position_embeddings=cfg["max_pos"], rms_norm_eps=cfg["eps"], pad_token_id=0, bos_token_id=1,
                     eos_token_id=2, tie_word_embeddings=False, head_dim=cfg["head_dim"])
    hc._attn_implementation = "eager"
    hc.rope_theta = cfg["rope_theta"]
    hc.rope_scaling = cfg.get("rope_scaling")            # None | {"type": "linear" | "dynamic", "factor": f}  (modeling_llama.py:431-456)
    hc.pretraining_tp = 1
    hc.attention_bias = False
    hc.attention_dropout = 0.0
    model = RefLM(hc).eval()
    sd = {"model.embed_tokens.weight": weights["embed"], "model.norm.weight": weights["norm"], "lm_head.weight": weights["lm_head"]}
    names = {"ln1": "input_layernorm", "ln2": "post_attention_layernorm", "wq": "self_attn.q_proj", "wk": "self_attn.k_proj",
             "wv": "self_attn.v_proj", "wo": "self_attn.o_proj", "wg": "mlp.gate_proj", "wu": "mlp.up_proj", "wd": "mlp.down_proj"}
    for i in range(cfg["layers"]):
        for k, v in names.items():
            sd[f"model.layers.{i}.{v}.weight"] = weights[f"layers.{i}.{k}"]
    missing, unexpected = model.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=False)
    assert not unexpected, unexpected
    assert all("rotary" in m or "inv_freq" in m for m in missing), missing
    model.generation_config = GenerationConfig(pad_token_id=0, eos_token_id=None, bos_token_id=1)   # EOS only when a run passes one
    return model


class Recorder:
    """Wraps the reference model's own methods (no edits) to record what each step is fed."""

    def __init__(self, model, capture_attn_steps=()):
        self.steps = []
        self.model = model
        self.capture_attn_steps = set(capture_attn_steps)
        self.attn = {}
        inner = model.model.LlamaModeljforward
        outer = model.jforward_multilevel

        def wrapped_inner(*a, **k):
            self._cur = {"ids": k["input_ids"][0].tolist(), "positions": k["position_ids"][0].tolist(),
                         "level_sizes": list(k["level_sizes"]), "lguess": len(k["guess"]) if k["guess"] is not None else 0,
                         "is_prefill": bool(k["is_prefill"]),
                         "P": 0 if k["past_key_values"] is None else int(k["past_key_values"][0][0].shape[2])}
            return inner(*a, **k)

        def wrapped_outer(*a, **k):
            step_no = len(self.steps)
            hooks = []
            cap = {}
            if step_no in self.capture_attn_steps:
                for li, layer in enumerate(model.model.layers):
                    at = layer.self_attn
                    for nm in ("q_proj", "k_proj", "v_proj"):
                        hooks.append(getattr(at, nm).register_forward_hook(
                            lambda mod, inp, out, li=li, nm=nm: cap.__setitem__(f"L{li}.{nm}", out[0].detach().clone())))
                    hooks.append(at.o_proj.register_forward_pre_hook(
                        lambda mod, inp, li=li: cap.__setitem__(f"L{li}.attn_out", inp[0][0].detach().clone())))
            out = outer(*a, **k)
            for h in hooks:
                h.remove()
            rec = self._cur
            rec["n_input"] = int(k["input_ids"].shape[1])
            rec["kvcache_len"] = int(out.kvcache_len)
            rec["step_len"] = int(out.step_len)
            rec["out_argmax"] = int(torch.argmax(out.out_logits, dim=-1).item())
            rec["inp_argmax"] = torch.argmax(out.inp_logits, dim=-1)[0].tolist()
            rec["guess_argmax"] = torch.argmax(out.guess_logits, dim=-1)[0].tolist() if rec["lguess"] > 0 else []
            if step_no in self.capture_attn_steps:
                for li in range(len(model.model.layers)):
                    cap[f"L{li}.K"] = out.past_key_values[li][0][0].detach().clone()
                    cap[f"L{li}.V"] = out.past_key_values[li][1][0].detach().clone()
                self.attn[step_no] = cap
            self.steps.append(rec)
            return out

        model.model.LlamaModeljforward = wrapped_inner
        model.jforward_multilevel = wrapped_outer

    def remove(self):
        del self.model.model.LlamaModeljforward
        del self.model.jforward_multilevel


def run_ref_greedy(model, prompt, W, N, G, max_length, seed, pool_from_prompt=0, eos=None, capture=()):
    D.CONFIG_MAP.clear()
    D.CONFIG_MAP.update(dict(WINDOW_SIZE=W, LEVEL=N, GUESS_SET_SIZE=G, ALWAYS_FWD_ONE=1, DEBUG=1, POOL_FROM_PROMPT=pool_from_prompt,
                             USE_FLASH=0, log=[]))
    rec = Recorder(model, capture)
    random.seed(seed)
    ids = torch.tensor([prompt])
    kw = {}
    if eos is not None:
        kw["eos_token_id"] = [eos]
    with torch.no_grad():
        out = D.jacobi_greedy_search_multilevel(model, ids, logits_processor=[], stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length)]),
                                                pad_token_id=0, attention_mask=torch.ones_like(ids), use_cache=True,
                                                return_dict_in_generate=False, output_attentions=False, output_hidden_states=False,
                                                output_scores=False, chat=False, **({"eos_token_id": [eos]} if eos is not None else {"eos_token_id": None}))
    rec.remove()
    gen, steps, _ = D.CONFIG_MAP["log"][-1]
    return out[0].tolist(), steps, gen, rec


PROMPTS = {
    "rep": [1, 5, 9, 17, 33, 5, 9, 17, 44, 5, 9],
    "rep2": [3, 7, 7, 3, 7, 7, 3, 11, 7, 7, 3, 7],
    "rnd": [17, 93, 41, 8, 120, 66, 5, 77, 101, 29, 54, 12, 99, 3, 81, 60],
}
MODELS = {"tiny-d16": dict(seed=0, std=0.02), "tiny-d64": dict(seed=1, std=0.05), "tiny-d128": dict(seed=2, std=0.05)}


def get_model(name):
    cfg = make_config(name, max_pos=512)
    w = random_weights_numpy(cfg, **MODELS[name])
    return cfg, w, build_ref_model(cfg, w)


def plain_greedy_hf(model, prompt, max_length):
    """Ordinary greedy decoding with the reference's model class (its full-causal path)."""
    ids = list(prompt)
    with torch.no_grad():
        while len(ids) < max_length:
            t = torch.tensor([ids])
            hidden = model.model.LlamaModeljforward(input_ids=t, attention_mask=torch.ones_like(t), position_ids=torch.arange(len(ids))[None],
                                                    past_key_values=None, use_cache=False, output_attentions=False, output_hidden_states=False,
                                                    return_dict=True, is_prefill=True, level_sizes=[len(ids) - 1], guess_size=2, guess=None,
                                                    use_flash=False)[0]
            ids.append(int(torch.argmax(model.lm_head(hidden[0, -1]).float()).item()))
    return ids


def gen_e2e_greedy():
    runs = []
    attn_npz = {}
    plan = [
        ("tiny-d16", "rep", 5, 4, 5, 48, (1, 2, 3), 0, None),
        ("tiny-d16", "rep2", 3, 3, 2, 40, (1,), 1, None),
        ("tiny-d64", "rep", 5, 3, 3, 40, (1, 2), 0, None),
        ("tiny-d64", "rep2", 7, 5, 7, 48, (1,), 1, None),
        ("tiny-d64", "rnd", 4, 4, 4, 40, (5,), 0, None),
        ("tiny-d128", "rep", 15, 5, 15, 40, (1,), 0, None),
        ("tiny-d128", "rep2", 6, 7, 6, 36, (2,), 0, None),
    ]
    for (mname, pname, W, N, G, new, seeds, pfp, eos) in plan:
        cfg, w, model = get_model(mname)
        prompt = [t % cfg["vocab"] for t in PROMPTS[pname]]
        plain = plain_greedy_hf(model, prompt, len(prompt) + new)
        for seed in seeds:
            capture = (2, 5, 9) if (mname, pname, seed) in (("tiny-d64", "rep", 1), ("tiny-d128", "rep", 1)) else ()
            toks, steps, gen, rec = run_ref_greedy(model, prompt, W, N, G, len(prompt) + new, seed, pfp, eos, capture)
            min_margin = None
            runs.append({"model": mname, "model_seed": MODELS[mname]["seed"], "std": MODELS[mname]["std"], "prompt": prompt, "W": W, "N": N, "G": G,
                         "max_length": len(prompt) + new, "seed": seed, "pool_from_prompt": pfp, "eos": eos, "tokens": toks,
                         "steps": steps, "generated": gen, "equals_plain_greedy": toks == plain[:len(toks)], "plain": plain, "trace": rec.steps})
            print(mname, pname, W, N, G, "seed", seed, "steps", steps, "gen", gen, "S", round(gen / steps, 2), "plain==", toks == plain[:len(toks)])
            for st, cap in rec.attn.items():
                for k, v in cap.items():
                    attn_npz[f"{mname}.{W}.{N}.{G}.s{seed}.step{st}.{k}"] = v.numpy()
    # EOS inside generation: pick a token that the rep run generates mid-way
    cfg, w, model = get_model("tiny-d16")
    prompt = PROMPTS["rep"]
    base = [r for r in runs if r["model"] == "tiny-d16" and r["seed"] == 1 and r["W"] == 5][0]
    eos = base["tokens"][len(prompt) + 20]
    toks, steps, gen, rec = run_ref_greedy(model, prompt, 5, 4, 5, len(prompt) + 48, 1, 0, eos)
    runs.append({"model": "tiny-d16", "model_seed": 0, "std": 0.02, "prompt": prompt, "W": 5, "N": 4, "G": 5, "max_length": len(prompt) + 48,
                 "seed": 1, "pool_from_prompt": 0, "eos": eos, "tokens": toks, "steps": steps, "generated": gen,
                 "equals_plain_greedy": None, "plain": None, "trace": rec.steps})
    print("eos run", eos, "len", len(toks), "steps", steps)
    dump("e2e_greedy.json", {"runs": runs})
    np.savez_compressed(os.path.join(GOLD, "attn_steps.npz"), **attn_npz)
    print("wrote attn_steps.npz", os.path.getsize(os.path.join(GOLD, "attn_steps.npz")))



def gen_e2e_greedy_wide():
    """BASELINE config 4's lookahead parameters (W = 20, N = 7, G = 20: steps of up to 6 * (20 + 20) = 240 tokens, six-token n-grams,
    the long-guess verification branch) on the tiny models, with and without POOL_FROM_PROMPT.  A separate fixture so that
    e2e_greedy.json stays byte-identical."""
    runs = []
    for (mname, pname, W, N, G, new, seed, pfp) in [("tiny-d128", "rep", 20, 7, 20, 48, 1, 1), ("tiny-d64", "rep2", 20, 7, 20, 40, 2, 0)]:
        cfg, w, model = get_model(mname)
        prompt = [t % cfg["vocab"] for t in PROMPTS[pname]]
        plain = plain_greedy_hf(model, prompt, len(prompt) + new)
        toks, steps, gen, rec = run_ref_greedy(model, prompt, W, N, G, len(prompt) + new, seed, pfp, None, ())
        runs.append({"model": mname, "model_seed": MODELS[mname]["seed"], "std": MODELS[mname]["std"], "prompt": prompt, "W": W, "N": N, "G": G,
                     "max_length": len(prompt) + new, "seed": seed, "pool_from_prompt": pfp, "eos": None, "tokens": toks,
                     "steps": steps, "generated": gen, "equals_plain_greedy": toks == plain[:len(toks)], "plain": plain, "trace": rec.steps})
        print("wide", mname, pname, W, N, G, "seed", seed, "steps", steps, "gen", gen, "S", round(gen / steps, 2), "plain==", toks == plain[:len(toks)],
              "max T", max(len(st["ids"]) for st in rec.steps))
    dump("e2e_greedy_wide.json", {"runs": runs})


def gen_e2e_unlimited():
    """GUESS_SET_SIZE = -1 ("unlimited" pool): both reference loops gate the verification branch on GUESS_SET_SIZE > 0
    (lade/decoding.py:402, :948), so the pool is only written - one token per step.  Small separate fixture."""
    runs = []
    for (mname, pname, W, N, new, seed, pfp) in [("tiny-d16", "rep", 5, 4, 32, 1, 0), ("tiny-d64", "rep2", 4, 3, 28, 2, 1)]:
        cfg, w, model = get_model(mname)
        prompt = [t % cfg["vocab"] for t in PROMPTS[pname]]
        toks, steps, gen, rec = run_ref_greedy(model, prompt, W, N, -1, len(prompt) + new, seed, pfp, None)
        runs.append({"model": mname, "model_seed": MODELS[mname]["seed"], "std": MODELS[mname]["std"], "prompt": prompt, "W": W, "N": N, "G": -1,
                     "max_length": len(prompt) + new, "seed": seed, "pool_from_prompt": pfp, "eos": None, "tokens": toks, "steps": steps,
                     "generated": gen, "trace": rec.steps})
        print("unlimited", mname, "steps", steps, "gen", gen)
    dump("e2e_unlimited.json", {"runs": runs})

def gen_e2e_dynamic_ntk():
    """rope_scaling = {"type": "dynamic"} (LlamaDynamicNTKScalingRotaryEmbedding, modeling_llama.py:292-318): max_position_embeddings far
    below the generated length, so the tables are rebuilt - with a new base - at nearly every step.  Greedy lookahead runs + the plain
    greedy stream of the same model (which rebuilds at different lengths: the two streams need not agree, and are not required to)."""
    runs = []
    for (mname, pname, W, N, G, new, seed, pfp, mp, factor) in [("tiny-d64", "rep", 5, 4, 5, 56, 1, 0, 24, 2.0), ("tiny-d16", "rnd", 4, 3, 4, 48, 2, 1, 32, 4.0)]:
        cfg = make_config(mname, max_pos=mp, rope_scaling={"type": "dynamic", "factor": factor})
        w = random_weights_numpy(cfg, **MODELS[mname])
        model = build_ref_model(cfg, w)
        prompt = [t % cfg["vocab"] for t in PROMPTS[pname]]
        toks, steps, gen, rec = run_ref_greedy(model, prompt, W, N, G, len(prompt) + new, seed, pfp, None, ())
        runs.append({"model": mname, "model_seed": MODELS[mname]["seed"], "std": MODELS[mname]["std"], "prompt": prompt, "W": W, "N": N, "G": G,
                     "max_length": len(prompt) + new, "seed": seed, "pool_from_prompt": pfp, "eos": None, "tokens": toks, "steps": steps,
                     "generated": gen, "max_pos": mp, "rope_scaling": cfg["rope_scaling"], "plain": None, "equals_plain_greedy": None, "trace": rec.steps})
        print("dynamic-ntk", mname, "max_pos", mp, "factor", factor, "steps", steps, "gen", gen, "S", round(gen / steps, 2),
              "longest step", max(st["step_len"] for st in rec.steps))
    dump("e2e_dynamic_ntk.json", {"runs": runs})

def gen_e2e_dynamic_ntk_again():
    """Two CONSECUTIVE generate() calls on ONE reference model under rope_scaling = dynamic: `max_seq_len_cached` and the rebuilt `inv_freq`
    live on the rotary module (modeling_llama.py:243-246, :299-316) and are never reset, so the second call starts from the LARGEST base the
    first one reached - its early steps rotate with other tables than a fresh model's would (round-4 advice: the port reset that state per
    sequence).  Call 2 is shorter than call 1, so it never rebuilds at all.  Also recorded: what a FRESH model yields for call 2."""
    runs = []
    # (weights drawn wider than the other fixtures' - std 0.25 / 0.3 - and a larger factor: the tiny models of MODELS are nearly insensitive to
    # the rotation base, and the point of this fixture is that the second call's tokens DIFFER from a fresh model's)
    for (mname, wseed, std, W, N, G, seed, mp, factor, p1, new1, p2, new2) in [("tiny-d64", 11, 0.25, 5, 4, 5, 1, 16, 8.0, "rep", 72, "rep2", 24),
                                                                              ("tiny-d16", 12, 0.3, 4, 3, 4, 2, 16, 8.0, "rnd", 64, "rep", 20)]:
        cfg = make_config(mname, max_pos=mp, rope_scaling={"type": "dynamic", "factor": factor})
        w = random_weights_numpy(cfg, seed=wseed, std=std)
        model = build_ref_model(cfg, w)
        calls = []
        for (pname, new) in ((p1, new1), (p2, new2)):
            prompt = [t % cfg["vocab"] for t in PROMPTS[pname]]
            toks, steps, gen, rec = run_ref_greedy(model, prompt, W, N, G, len(prompt) + new, seed, 0, None, ())
            calls.append({"prompt": prompt, "max_length": len(prompt) + new, "tokens": toks, "steps": steps, "generated": gen,
                          "longest_step": max(st["step_len"] for st in rec.steps)})
        fresh = build_ref_model(cfg, w)
        prompt2 = calls[1]["prompt"]
        toks_f, steps_f, _g, _r = run_ref_greedy(fresh, prompt2, W, N, G, calls[1]["max_length"], seed, 0, None, ())
        runs.append({"model": mname, "model_seed": wseed, "std": std, "W": W, "N": N, "G": G, "seed": seed, "max_pos": mp,
                     "rope_scaling": cfg["rope_scaling"], "calls": calls, "second_call_on_a_fresh_model": {"tokens": toks_f, "steps": steps_f}})
        print("dynamic-ntk again", mname, [(c["steps"], c["generated"], c["longest_step"]) for c in calls], "fresh-model call 2 differs:", toks_f != calls[1]["tokens"])
    dump("e2e_dynamic_ntk_again.json", {"runs": runs})

# ------------------------------------------------------------------ sampling


def gen_e2e_sample():
    runs = []
    plan = [("tiny-d16", "rep", 5, 4, 5, 40, 1, dict(temperature=0.8)),
            ("tiny-d16", "rep", 5, 4, 5, 40, 2, dict(temperature=0.3)),
            ("tiny-d64", "rep2", 4, 3, 4, 36, 3, dict(temperature=0.5, top_k=20)),
            ("tiny-d64", "rep", 5, 3, 3, 36, 4, dict(temperature=0.4, top_p=0.9)),
            ("tiny-d16", "rep2", 3, 3, 2, 30, 5, dict(temperature=0.2)),
            ("tiny-d16", "rep", 5, 4, 5, 40, 6, dict(temperature=0.05))]
    for (mname, pname, W, N, G, new, seed, wk) in plan:
        cfg, w, model = get_model(mname)
        prompt = [t % cfg["vocab"] for t in PROMPTS[pname]]
        warp = LogitsProcessorList()
        if "temperature" in wk:
            warp.append(TemperatureLogitsWarper(wk["temperature"]))
        if "top_k" in wk:
            warp.append(TopKLogitsWarper(wk["top_k"]))
        if "top_p" in wk:
            warp.append(TopPLogitsWarper(wk["top_p"]))
        D.CONFIG_MAP.clear()
        D.CONFIG_MAP.update(dict(WINDOW_SIZE=W, LEVEL=N, GUESS_SET_SIZE=G, ALWAYS_FWD_ONE=1, DEBUG=1, POOL_FROM_PROMPT=0, USE_FLASH=0, log=[]))
        rec = Recorder(model)
        random.seed(seed)
        torch.manual_seed(seed)
        ids = torch.tensor([prompt])
        with torch.no_grad():
            out = D.jacobi_sample_multilevel(model, ids, logits_processor=LogitsProcessorList(), logits_warper=warp,
                                             stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(len(prompt) + new)]),
                                             pad_token_id=0, eos_token_id=None, attention_mask=torch.ones_like(ids), use_cache=True,
                                             return_dict_in_generate=False, output_attentions=False, output_hidden_states=False,
                                             output_scores=False, chat=False)
        rec.remove()
        gen, steps, _ = D.CONFIG_MAP["log"][-1]
        runs.append({"model": mname, "model_seed": MODELS[mname]["seed"], "std": MODELS[mname]["std"], "prompt": prompt, "W": W, "N": N, "G": G,
                     "max_length": len(prompt) + new, "seed": seed, "warp": wk, "tokens": out[0].tolist(), "steps": steps, "generated": gen,
                     "trace": rec.steps})
        print("sample", mname, wk, "steps", steps, "gen", gen)
    dump("e2e_sample.json", {"runs": runs})



def gen_e2e_sample_eos():
    """Sampling runs that exercise filter_window (EOS replaced in the newest window level, lade/decoding.py:578-580), the EOS
    stop of the sampling loop (:592-604) and POOL_FROM_PROMPT under sampling (:367-368, :602-603)."""
    runs = []
    for (mname, pname, W, N, G, new, seed, wk, pfp, eos_rank) in [("tiny-d16", "rep", 5, 4, 5, 40, 1, dict(temperature=0.8), 0, 0),
                                                                 ("tiny-d16", "rep", 5, 4, 5, 40, 2, dict(temperature=0.3), 1, 1),
                                                                 ("tiny-d64", "rep2", 4, 3, 4, 36, 3, dict(temperature=0.5, top_k=20), 1, None)]:
        cfg, w, model = get_model(mname)
        prompt = [t % cfg["vocab"] for t in PROMPTS[pname]]
        warp = LogitsProcessorList()
        if "temperature" in wk:
            warp.append(TemperatureLogitsWarper(wk["temperature"]))
        if "top_k" in wk:
            warp.append(TopKLogitsWarper(wk["top_k"]))

        def run(eos):
            D.CONFIG_MAP.clear()
            D.CONFIG_MAP.update(dict(WINDOW_SIZE=W, LEVEL=N, GUESS_SET_SIZE=G, ALWAYS_FWD_ONE=1, DEBUG=1, POOL_FROM_PROMPT=pfp, USE_FLASH=0, log=[]))
            rec = Recorder(model)
            random.seed(seed)
            torch.manual_seed(seed)
            ids = torch.tensor([prompt])
            with torch.no_grad():
                out = D.jacobi_sample_multilevel(model, ids, logits_processor=LogitsProcessorList(), logits_warper=warp,
                                                 stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(len(prompt) + new)]),
                                                 pad_token_id=0, eos_token_id=None if eos is None else [eos], attention_mask=torch.ones_like(ids),
                                                 use_cache=True, return_dict_in_generate=False, output_attentions=False, output_hidden_states=False,
                                                 output_scores=False, chat=False)
            rec.remove()
            gen, steps, _ = D.CONFIG_MAP["log"][-1]
            return out[0].tolist(), steps, gen, rec

        eos = None
        if eos_rank is not None:                     # an EOS id that the free run does produce: the run must stop there
            base, _, _, _ = run(None)
            gen_toks = base[len(prompt) + 6:]
            import collections
            eos = collections.Counter(gen_toks).most_common(eos_rank + 1)[eos_rank][0]
        toks, steps, gen, rec = run(eos)
        runs.append({"model": mname, "model_seed": MODELS[mname]["seed"], "std": MODELS[mname]["std"], "prompt": prompt, "W": W, "N": N, "G": G,
                     "max_length": len(prompt) + new, "seed": seed, "warp": wk, "pool_from_prompt": pfp, "eos": eos, "tokens": toks, "steps": steps,
                     "generated": gen, "trace": rec.steps})
        print("sample-eos", mname, wk, "pfp", pfp, "eos", eos, "steps", steps, "gen", gen, "len", len(toks) - len(prompt))
    dump("e2e_sample_eos.json", {"runs": runs})

# ------------------------------------------------------------------ lookahead parallel (gloo)


def _lp_worker(rank, R, port, mname, prompt, W, N, G, max_length, seed, q, pfp=0, eos=None):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=R)
    torch.set_num_threads(1)
    cfg, w, model = get_model(mname)
    D.CONFIG_MAP.clear()
    D.CONFIG_MAP.update(dict(WINDOW_SIZE=W, LEVEL=N, GUESS_SET_SIZE=G, ALWAYS_FWD_ONE=1, DEBUG=1, POOL_FROM_PROMPT=pfp, USE_FLASH=0, log=[],
                             DIST_WORKERS=R, LOCAL_RANK=rank))
    rec = Recorder(model)
    random.seed(seed + rank * 1000)      # ranks differ: the reference broadcasts rank 0's window (:906)
    ids = torch.tensor([prompt])
    with torch.no_grad():
        out = D.jacobi_greedy_search_multilevel(model, ids, logits_processor=[], stopping_criteria=StoppingCriteriaList([MaxLengthCriteria(max_length)]),
                                                pad_token_id=0, eos_token_id=None if eos is None else [eos], attention_mask=torch.ones_like(ids), use_cache=True,
                                                return_dict_in_generate=False, output_attentions=False, output_hidden_states=False,
                                                output_scores=False, chat=False)
    steps = len(rec.steps)
    q.put((rank, out[0].tolist(), steps, rec.steps))
    dist.barrier()
    dist.destroy_process_group()


def gen_e2e_lp():
    import torch.multiprocessing as mp
    runs = []
    port = 29611
    eos_of = {}
    for (mname, pname, W, N, G, new, seed, R, pfp) in [("tiny-d16", "rep", 5, 4, 5, 48, 1, 2, 0), ("tiny-d16", "rep", 5, 4, 5, 48, 1, 3, 0),
                                                       ("tiny-d64", "rep2", 7, 5, 7, 40, 1, 2, 0), ("tiny-d64", "rep", 5, 3, 3, 40, 2, 4, 0),
                                                       ("tiny-d16", "rep", 5, 4, 5, 48, 3, 2, 1), ("tiny-d64", "rep2", 6, 4, 4, 40, 2, 3, 1),
                                                       ("tiny-d64", "rep", 15, 5, 15, 40, 1, 8, 0),     # BASELINE config 5's W/N/G over 8 ranks
                                                       ("tiny-d16", "rep", 5, 4, 5, 48, 1, 2, "eos")]:  # the first run again, stopping at an EOS it generates
        cfg = make_config(mname)
        prompt = [t % cfg["vocab"] for t in PROMPTS[pname]]
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port += 1
        eos = None
        if pfp == "eos":
            pfp, eos = 0, eos_of[(mname, pname, W, N, G, seed)][len(prompt) + 20]
        procs = [ctx.Process(target=_lp_worker, args=(r, R, port, mname, prompt, W, N, G, len(prompt) + new, seed, q, pfp, eos)) for r in range(R)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=600) for _ in range(R)])
        for p in procs:
            p.join()
        assert all(r[1] == res[0][1] for r in res)
        eos_of.setdefault((mname, pname, W, N, G, seed), res[0][1])
        runs.append({"model": mname, "model_seed": MODELS[mname]["seed"], "std": MODELS[mname]["std"], "prompt": prompt, "W": W, "N": N, "G": G,
                     "max_length": len(prompt) + new, "seed": seed, "R": R, "pool_from_prompt": pfp, "eos": eos, "tokens": res[0][1], "steps": res[0][2],
                     "rank_traces": [r[3] for r in res]})
        print("LP", mname, "R", R, "pool_from_prompt", pfp, "steps", res[0][2])
    dump("e2e_lp.json", {"runs": runs})


if __name__ == "__main__":
    what = sys.argv[1:] or ["pool", "mask", "greedy", "sample", "lp", "unlimited", "sample_eos"]
    torch.set_num_threads(4)
    if "pool" in what:
        gen_pool()
    if "mask" in what:
        gen_masks()
    if "greedy" in what:
        gen_e2e_greedy()
    if "sample" in what:
        gen_e2e_sample()
    if "lp" in what:
        gen_e2e_lp()
    if "unlimited" in what:
        gen_e2e_unlimited()
    if "greedy_wide" in what:          # not in the default list: added in round 2 without regenerating the other fixtures
        gen_e2e_greedy_wide()
    if "sample_eos" in what:
        gen_e2e_sample_eos()
    if "dynamic_ntk" in what:          # round 4: a separate fixture, the others stay byte-identical
        gen_e2e_dynamic_ntk()
    if "dynamic_ntk_again" in what:    # round 5: two consecutive calls on one model
        gen_e2e_dynamic_ntk_again()
