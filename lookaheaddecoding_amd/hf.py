"""Glue between HuggingFace `generate()` and the MI355X step engine: the patched entry points.

Reference: `greedy_search_proxy` / `sample_proxy` (lade/decoding.py:15-34), the signatures of
`jacobi_greedy_search_multilevel` (:697-715) and `jacobi_sample_multilevel` (:137-154), and the way
`augment_generate` swaps them into `GenerationMixin` (lade/utils.py:62-66).  The reference targets
transformers 4.36.2 (`GenerationMixin.greedy_search` / `.sample`); from 4.4x on both became
`GenerationMixin._sample(..., generation_config, ...)` with `generation_config.do_sample` selecting the
mode - both hook points are patched when present.

The reference also overwrites every method of HF's Llama classes with its own copy
(lade/utils.py:40-56).  Here the model step does not run through HF modules at all: the first lookahead
`generate()` call on a LlamaForCausalLM builds a `StepEngine` over the module's own weight tensors
(fused QKV / gate-up copies are made once) and caches it on the model object.
"""
from __future__ import annotations

import os
import random
from typing import Optional

import torch

from . import cabi
from .decoding import CONFIG_MAP, FUNC_MAP, LookaheadDecoder
from .engine import StepEngine

_ENGINE_ATTR = "_lade_engine"


def _rope_params(c):
    """(theta, scaling dict or None) from either config generation (4.36: rope_theta / rope_scaling; 5.x: rope_parameters)."""
    rp = getattr(c, "rope_parameters", None)
    rp = dict(rp) if isinstance(rp, dict) else {}
    theta = getattr(c, "rope_theta", None)
    if theta is None:
        theta = rp.get("rope_theta", 10000.0)
    scaling = getattr(c, "rope_scaling", None)
    if scaling is None and rp.get("rope_type", rp.get("type", "default")) not in (None, "default"):
        scaling = rp
    if isinstance(scaling, dict) and scaling.get("rope_type", scaling.get("type", "default")) in (None, "default"):
        scaling = None
    return float(theta), scaling


def config_from_hf(model) -> dict:
    """Engine config of an HF LlamaForCausalLM.  Anything the HIP step does not implement fails loudly here - the reference
    only patches the Llama classes (lade/utils.py:40-56), so any other architecture raises there too."""
    c = model.config
    if type(model).__name__ != "LlamaForCausalLM" or getattr(c, "model_type", "llama") != "llama":
        raise cabi.LadeHipError(f"lookahead decoding supports LlamaForCausalLM only (got {type(model).__name__}, model_type="
                                f"{getattr(c, 'model_type', None)!r}); the reference patches the Llama classes only (lade/utils.py:40-56)")
    theta, scaling = _rope_params(c)
    if scaling is not None:
        kind = scaling.get("rope_type", scaling.get("type"))
        if kind not in ("linear", "llama3", "dynamic"):
            # ('dynamic' = LlamaDynamicNTKScalingRotaryEmbedding, lade/models/modeling_llama.py:292-318: per-step rows, engine._ntk_rows)
            raise cabi.LadeHipError(f"rope_scaling type {kind!r} is not implemented by the HIP step (default, linear, dynamic and llama3 are)")
    if getattr(c, "attention_bias", False) or getattr(c, "mlp_bias", False):
        raise cabi.LadeHipError("projection biases are not supported (Llama-2 has none)")
    if getattr(c, "pretraining_tp", 1) not in (None, 1):
        raise cabi.LadeHipError("pretraining_tp > 1 is out of scope (DESIGN.md section 8)")
    heads = c.num_attention_heads
    head_dim = getattr(c, "head_dim", None) or c.hidden_size // heads
    return dict(hidden=c.hidden_size, inter=c.intermediate_size, layers=c.num_hidden_layers, heads=heads,
                kv_heads=getattr(c, "num_key_value_heads", heads) or heads, head_dim=head_dim, vocab=c.vocab_size,
                eps=float(c.rms_norm_eps), rope_theta=theta, rope_scaling=scaling,
                max_pos=int(getattr(c, "max_position_embeddings", 4096)))


def weights_from_hf(model) -> dict:
    """References (no copies) to the HF module's parameters in the engine's naming."""
    m = model.model
    w = {"embed": m.embed_tokens.weight.data, "norm": m.norm.weight.data, "lm_head": model.lm_head.weight.data}
    for i, layer in enumerate(m.layers):
        a, mlp = layer.self_attn, layer.mlp
        for bias_owner in (a.q_proj, a.k_proj, a.v_proj, a.o_proj, mlp.gate_proj, mlp.up_proj, mlp.down_proj):
            if getattr(bias_owner, "bias", None) is not None:
                raise cabi.LadeHipError("projection biases are not supported (Llama has none)")
        p = f"layers.{i}."
        w.update({p + "ln1": layer.input_layernorm.weight.data, p + "ln2": layer.post_attention_layernorm.weight.data,
                  p + "wq": a.q_proj.weight.data, p + "wk": a.k_proj.weight.data, p + "wv": a.v_proj.weight.data,
                  p + "wo": a.o_proj.weight.data, p + "wg": mlp.gate_proj.weight.data, p + "wu": mlp.up_proj.weight.data,
                  p + "wd": mlp.down_proj.weight.data})
    return w


def get_engine(model, need_seq: int, need_T: int, keep_rows: int = 0) -> StepEngine:
    """The model's cached StepEngine.  Built once over the module's own weight tensors; when a later call needs a longer
    KV cache or a wider step, the cache / workspaces grow in place (`StepEngine.grow`: the first keep_rows cache rows
    survive, so an `EngineCache` a caller still holds stays valid) - the fused weight copies are never duplicated."""
    eng: Optional[StepEngine] = getattr(model, _ENGINE_ATTR, None)
    if eng is not None:
        if eng.S_max < need_seq or eng.max_T < need_T:
            eng.grow(max(need_seq, 2 * eng.S_max if eng.S_max < need_seq else 0), need_T, keep_rows)
        return eng
    dev = model.lm_head.weight.device
    if dev.type != "cuda":
        raise cabi.LadeHipError("USE_LADE=1 needs the model on a GPU (the HIP hot path has no CPU fallback)")
    dtype = model.lm_head.weight.dtype
    cfg = config_from_hf(model)
    # sized for the model's whole context window when that is affordable (<= 8 GiB of K/V), otherwise for what this call needs
    kv_bytes_per_row = 2 * cfg["layers"] * cfg["kv_heads"] * cfg["head_dim"] * model.lm_head.weight.element_size()
    full = cfg["max_pos"] if cfg["max_pos"] * kv_bytes_per_row <= (8 << 30) else 0
    # step width 2048: long prompts prefill in few, wide causal chunks (the library GEMMs run ~30 % faster at 2048 rows than at 512)
    eng = StepEngine(cfg, weights_from_hf(model), dtype=dtype, device=dev, max_seq=max(need_seq, 2048, full), max_T=max(need_T, 2048))
    setattr(model, _ENGINE_ATTR, eng)
    return eng


# ---- model-step boundary: jforward_multilevel on the HIP step engine ---------------------------

class EngineCache:
    """Stands in for the reference's `past_key_values` tuple: the K/V rows live preallocated inside the step engine
    (K row-major, V transposed); this handle only knows how many rows are valid.  The reference's caller edits the
    cache after a step (copy the accepted candidate rows down, then slice; lade/decoding.py:1148-1163) - the same two
    operations are `move_rows` and `crop` here."""

    def __init__(self, engine: StepEngine, length: int):
        self.engine, self.length = engine, int(length)

    def __len__(self) -> int:
        return self.length

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.length

    def crop(self, length: int) -> "EngineCache":
        assert 0 <= length <= self.length
        self.length = int(length)
        return self

    def move_rows(self, src: int, dst: int, cnt: int) -> "EngineCache":
        from . import ops
        if cnt > 0 and src != dst:
            ops.kv_commit(self.engine.kv, src, dst, cnt)
        return self


class StepOutput:
    """What the reference's loop reads from `jforward_multilevel`'s return value (lade/decoding.py:1012-1102):
    out_logits [1,V], inp_logits [1,window,V], guess_logits [1,lguess,V] (fp32), past_key_values, kvcache_len, step_len.
    `logits` is None: lm_head runs only on the rows above (the reference runs it on all T rows, modeling_llama.py:1541)."""

    def __init__(self):
        self.logits = None
        self.out_logits = self.inp_logits = self.guess_logits = None
        self.past_key_values = None
        self.kvcache_len = self.step_len = 0
        self.loss = self.hidden_states = self.attentions = None


def jforward_multilevel(self, input_ids=None, past_tokens=None, guess_tokens=None, guess_size=2, not_seq=False, continue_all=False, level=3,
                        fill_level=-1, WINDOWS_SIZE=-1, dist_workers=-1, local_rank=-1, la_mask_offset=0, use_flash=False, attention_mask=None,
                        position_ids=None, past_key_values=None, inputs_embeds=None, labels=None, use_cache=None, output_attentions=None,
                        output_hidden_states=None, return_dict=None) -> StepOutput:
    """One lookahead model step with the reference's signature (lade/models/modeling_llama.py:1381-1608) on the HIP
    step engine: assembles [inputs | L0 | L1.. | candidates] and their positions (:1487-1511), runs the fused step
    under the closed-form mask, and returns the three logit slices the decode loop reads.  `past_tokens` may already be
    a lookahead-parallel shard (lade/decoding.py:973-986): level sizes and the mask offsets follow from the list
    lengths exactly as in the reference.  `past_key_values`: None (start a sequence) or the EngineCache a previous
    call returned.  use_flash / not_seq / la_mask_offset are accepted for signature compatibility; there is one
    attention path."""
    from .ops import StepMask
    assert labels is None, " Inference Mode "
    assert input_ids.size(0) == 1, " single batch only "
    assert continue_all is False or continue_all == 0
    assert fill_level != -1
    if level is not None:
        assert level == len(past_tokens) + 1
        assert guess_size == level - 1
    gs = guess_size
    n_input = input_ids.size(1)
    in_ids = [int(t) for t in input_ids[0].tolist()]
    in_pos = [int(t) for t in position_ids[0].tolist()]
    lst_id = in_pos[-1]
    past_size = 0 if past_key_values is None else len(past_key_values)
    level_sizes, lv_ids, lv_pos = [], [], []
    for ll in range(fill_level + 1):
        toks = [int(t) for t in past_tokens[ll]]
        level_sizes.append(len(toks))
        lv_ids += toks
        if ll == 0:
            lv_pos += list(range(lst_id + 1, lst_id + 1 + len(toks)))
        else:
            off = len(past_tokens[0]) + 1 - len(toks)
            lv_pos += list(range(lst_id + ll + off, lst_id + ll + off + len(toks)))
    g_ids = [int(t) for t in guess_tokens] if guess_tokens is not None else []
    lguess = len(g_ids)
    g_pos = list(range(lst_id + 1, lst_id + 1 + gs)) * (lguess // gs) if lguess else []
    ids, pos = in_ids + lv_ids + g_ids, in_pos + lv_pos + g_pos
    T = len(ids)
    is_prefill = past_tokens[1] is None
    window = level_sizes[fill_level]
    # a prefill is chunked by the engine; any other step must fit one forward (the reference default W=60 N=8 G=60 feeds 847 rows)
    eng = get_engine(self, past_size + T + 64, 512 if is_prefill else max(T, 512), keep_rows=past_size)
    if past_key_values is None:
        eng.reset()
    elif past_key_values.engine is not eng:
        raise cabi.LadeHipError("past_key_values belongs to another model's step engine")
    dev = eng.device
    rows = [n_input - 1] + list(range(T - lguess - window, T - lguess)) + list(range(T - lguess, T))
    if is_prefill:                                    # prompt (+ first window level): causal chunks over the growing cache
        logits, _ = eng.prefill(ids, rows, P0=past_size, pos=pos)
        logits = logits.float()
    else:
        if T > eng.max_T:
            raise cabi.LadeHipError(f"step of {T} tokens exceeds the engine's max_T={eng.max_T}")
        logits = eng.forward(torch.tensor(ids, dtype=torch.int32, device=dev), torch.tensor(pos, dtype=torch.int32, device=dev),
                             StepMask.from_levels(n_input, level_sizes, lguess, gs, past_size),
                             torch.tensor(rows, dtype=torch.int32, device=dev), len(rows)).float()
    ret = StepOutput()
    ret.out_logits = logits[0:1]
    ret.inp_logits = logits[1:1 + window].unsqueeze(0)
    if lguess:
        ret.guess_logits = logits[1 + window:].unsqueeze(0)
    ret.past_key_values = EngineCache(eng, past_size + T)
    ret.kvcache_len = n_input + past_size
    ret.step_len = (attention_mask.size(1) if attention_mask is not None else past_size + n_input) + sum(level_sizes) + lguess
    return ret


def _criteria_limits(stopping_criteria, generation_config=None):
    max_length, eos = None, None
    for c in (stopping_criteria or []):
        if hasattr(c, "max_length") and c.max_length is not None:
            max_length = int(c.max_length) if max_length is None else min(max_length, int(c.max_length))
        if hasattr(c, "eos_token_id") and eos is None:
            e = c.eos_token_id
            e = e.tolist() if hasattr(e, "tolist") else e
            eos = e[0] if isinstance(e, (list, tuple)) else e
    return max_length, eos


def _run(self, input_ids, do_sample, warp, stopping_criteria, eos_token_id, generation_config, streamer, chat):
    W = CONFIG_MAP.get("WINDOW_SIZE", 60)            # reference defaults when config_lade was never called
    G = CONFIG_MAP.get("GUESS_SET_SIZE", 60)         # (lade/decoding.py:854-857)
    N = CONFIG_MAP.get("LEVEL", 8)
    assert CONFIG_MAP.get("ALWAYS_FWD_ONE", 1) == 1  # :873
    assert input_ids.size(0) == 1, " single batch only "      # modeling_llama.py:1448
    R = CONFIG_MAP.get("DIST_WORKERS", 1)
    max_length, eos_c = _criteria_limits(stopping_criteria, generation_config)
    if isinstance(eos_token_id, (list, tuple)):
        eos_token_id = eos_token_id[0] if eos_token_id else None
    if eos_token_id is None:
        eos_token_id = eos_c
    if max_length is None:
        max_length = int(getattr(generation_config, "max_length", 0) or 0) or input_ids.shape[1] + 20
    prompt = input_ids[0].tolist()
    gs = N - 1
    need_T = max((N - 1) * (W + G) + gs, 64)
    eng = get_engine(self, max_length + need_T + W + N + 64, need_T)
    key = (W, N, G, bool(CONFIG_MAP.get("POOL_FROM_PROMPT", 0)), R, bool(CONFIG_MAP.get("FORCE_LP", 0)))
    dec = getattr(self, "_lade_decoder", None)
    if dec is None or getattr(dec, "_key", None) != key or dec.e is not eng:
        lp = None
        force_lp = bool(CONFIG_MAP.get("FORCE_LP", 0))       # one rank on the lookahead-parallel path (its collective: a device copy, or a 1-rank RCCL all-gather in a joined group / with LADE_LP_COLLECTIVE=abi)
        if R > 1 or force_lp:
            from .parallel import LPContext
            lp = LPContext(rank=CONFIG_MAP.get("LOCAL_RANK", 0), world=R, force=force_lp)
        # under lookahead parallelism the rank-local part of a steady step can replay as a hipGraph segment (parallel.HipLPBackend).
        # With more than one rank that path has only ever run as threads / gloo processes sharing one GPU (no multi-GPU box was
        # available to any round): it stays opt-in (LADE_LP_GRAPH=1) until it has run on separate devices; the eager LP step is
        # GPU bound anyway (4.55 vs 4.59 ms at the 7B shape).  One rank (and plain decoding) replays graphs by default.
        graph = bool(int(os.environ.get("LADE_GRAPH", "1"))) and (R == 1 or bool(int(os.environ.get("LADE_LP_GRAPH", "0"))))
        dec = LookaheadDecoder(eng, W, N, G, pool_from_prompt=key[3], lp=lp, use_graph=graph)
        dec._key = key
        self._lade_decoder = dec
    # per-step output like the reference: decoded text printed incrementally under CHAT=1 (lade/decoding.py:1179-1195),
    # accepted tokens handed to the HF streamer as they are accepted (:1199-1200)
    rank0 = CONFIG_MAP.get("LOCAL_RANK", 0) == 0
    tok = getattr(self, "tokenizer", None) if (chat and rank0) else None
    shown = {"ids": [], "n": 0}

    def on_step(accepted):
        if not accepted:
            return
        if streamer is not None:
            streamer.put(torch.tensor(accepted, dtype=input_ids.dtype))
        if tok is not None:
            shown["ids"] += accepted
            text = tok.decode(shown["ids"], skip_special_tokens=True, spaces_between_special_tokens=False, clean_up_tokenization_spaces=True)
            print(text[shown["n"]:], flush=True, end="")
            shown["n"] = len(text)

    cb = on_step if (streamer is not None or tok is not None) else None
    if do_sample:
        out = dec.sample(prompt, max_length, warp=warp, eos_token_id=eos_token_id, rng=random, on_step=cb)
    else:
        out = dec.greedy(prompt, max_length, eos_token_id=eos_token_id, rng=random, on_step=cb)
    if CONFIG_MAP.get("DEBUG", 0) and rank0:
        print("\n==========================ACCELERATION===SUMMARY======================================")
        print("Generated tokens: ", out.generated, "Total steps: ", out.steps, " Compression ratio: ", round(out.generated / max(out.steps, 1), 2))
        print("======================================================================================", end="")
    res = torch.tensor([out.tokens], dtype=input_ids.dtype, device=input_ids.device)
    if streamer is not None:
        streamer.end()
    return res


def jacobi_greedy_search_multilevel(self, input_ids, logits_processor=None, stopping_criteria=None, max_length=None, pad_token_id=None,
                                    eos_token_id=None, output_attentions=None, output_hidden_states=None, output_scores=None,
                                    return_dict_in_generate=None, synced_gpus=False, streamer=None, chat=False, stop_token=None,
                                    generation_config=None, **model_kwargs):
    """Same call contract as lade/decoding.py:697-715 (extras `chat`, `stop_token`); returns LongTensor [1, len]."""
    assert not return_dict_in_generate, "return_dict_in_generate is not supported (lade/decoding.py:967)"
    assert logits_processor is None or len(logits_processor) == 0, "logits processors are not supported in greedy lookahead (:968)"
    if max_length is not None:
        from transformers.generation.stopping_criteria import MaxLengthCriteria, StoppingCriteriaList
        stopping_criteria = StoppingCriteriaList(list(stopping_criteria or []) + [MaxLengthCriteria(max_length)])
    return _run(self, input_ids, False, None, stopping_criteria, eos_token_id, generation_config, streamer, chat)


def _canonical_warper(warpers, T_cls, K_cls, P_cls):
    """(temperature, top_k, top_p) when the list is what `generate()` builds - at most one warper of each kind, in the order
    temperature, top-k, top-p, filtering with -inf and keeping one token - else None (the list is then applied as given)."""
    order = {T_cls: 0, K_cls: 1, P_cls: 2}
    kinds = [order[type(wp)] for wp in warpers]
    if kinds != sorted(set(kinds)):
        return None
    t, k, p = 1.0, 0, 1.0
    for wp in warpers:
        if type(wp) is T_cls:
            t = float(wp.temperature)
        else:
            if getattr(wp, "filter_value", -float("inf")) != -float("inf") or getattr(wp, "min_tokens_to_keep", 1) != 1:
                return None
            if type(wp) is K_cls:
                k = int(wp.top_k)
            else:
                p = float(wp.top_p)
    return t, k, p


def jacobi_sample_multilevel(self, input_ids, logits_processor=None, stopping_criteria=None, logits_warper=None, max_length=None,
                             pad_token_id=None, eos_token_id=None, output_attentions=None, output_hidden_states=None, output_scores=None,
                             return_dict_in_generate=None, synced_gpus=False, streamer=None, chat=False, generation_config=None,
                             **model_kwargs):
    """Same call contract as lade/decoding.py:137-154.  Warpers are restricted like the reference (:375-377)."""
    assert not return_dict_in_generate
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
    warpers = list(logits_warper or [])
    for wp in warpers:
        assert type(wp) in (TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper), f"please set top_k=0.0 and top_p=1.0 {wp}"

    from .sampling import Warper
    canonical = _canonical_warper(warpers, TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper)
    if canonical is not None:
        # HF's own construction order (temperature -> top-k -> top-p, -inf filter, one token kept): the device kernels apply it -
        # temperature alone inside the probability kernels (BASELINE config 3), with top-k / top-p as one lade_warp_rows launch
        warp = Warper(*canonical)
    else:
        def warp(scores):
            for wp in warpers:
                scores = wp(input_ids, scores)
            return scores

    return _run(self, input_ids, True, warp, stopping_criteria, eos_token_id, generation_config, streamer, chat)


def greedy_search_proxy(self, *args, **kwargs):
    """lade/decoding.py:15-26"""
    USE_LADE = int(os.environ.get("USE_LADE", 0))
    CHAT = int(os.environ.get("CHAT", 0))
    if USE_LADE:
        return jacobi_greedy_search_multilevel(self, *args, chat=bool(CHAT), **kwargs)
    return FUNC_MAP["greedy_search"](self, *args, **kwargs)


def sample_proxy(self, *args, **kwargs):
    """lade/decoding.py:28-34 (with USE_LADE=0 the reference dispatches to the saved *greedy* function, Appendix B.4;
    here the saved sample function is used)."""
    USE_LADE = int(os.environ.get("USE_LADE", 0))
    if USE_LADE:
        return jacobi_sample_multilevel(self, *args, chat=bool(int(os.environ.get("CHAT", 0))), **kwargs)
    return FUNC_MAP["sample"](self, *args, **kwargs)


def _sample_proxy(self, input_ids, logits_processor=None, stopping_criteria=None, generation_config=None, synced_gpus=False,
                  streamer=None, **model_kwargs):
    """transformers >= 4.4x: greedy and sampling both arrive here (`generation_config.do_sample`)."""
    USE_LADE = int(os.environ.get("USE_LADE", 0))
    if not USE_LADE:
        return FUNC_MAP["_sample"](self, input_ids, logits_processor=logits_processor, stopping_criteria=stopping_criteria,
                                   generation_config=generation_config, synced_gpus=synced_gpus, streamer=streamer, **model_kwargs)
    chat = bool(int(os.environ.get("CHAT", 0)))
    assert not generation_config.return_dict_in_generate, "return_dict_in_generate is not supported"
    if generation_config.do_sample:
        from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper
        warpers = [p for p in (logits_processor or [])]
        for wp in warpers:
            assert type(wp) in (TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper), f"please set top_k=0.0 and top_p=1.0 {wp}"
        return jacobi_sample_multilevel(self, input_ids, stopping_criteria=stopping_criteria, logits_warper=warpers, streamer=streamer,
                                        chat=chat, generation_config=generation_config)
    return jacobi_greedy_search_multilevel(self, input_ids, logits_processor=logits_processor, stopping_criteria=stopping_criteria,
                                           streamer=streamer, chat=chat, generation_config=generation_config)
