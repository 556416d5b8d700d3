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


def config_from_hf(model) -> dict:
    c = model.config
    heads = c.num_attention_heads
    head_dim = getattr(c, "head_dim", None) or c.hidden_size // heads
    theta = getattr(c, "rope_theta", None)
    if theta is None:
        rp = getattr(c, "rope_parameters", None) or {}
        theta = rp.get("rope_theta", 10000.0) if isinstance(rp, dict) else 10000.0
    return dict(hidden=c.hidden_size, inter=c.intermediate_size, layers=c.num_hidden_layers, heads=heads,
                kv_heads=getattr(c, "num_key_value_heads", heads) or heads, head_dim=head_dim, vocab=c.vocab_size,
                eps=float(c.rms_norm_eps), rope_theta=float(theta), max_pos=int(getattr(c, "max_position_embeddings", 4096)))


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


def get_engine(model, need_seq: int, need_T: int) -> StepEngine:
    """The model's cached StepEngine, rebuilt when the KV capacity or the step width must grow."""
    if type(model).__name__ not in ("LlamaForCausalLM",) and not hasattr(model, "model"):
        raise cabi.LadeHipError(f"lookahead decoding supports Llama-family models (got {type(model).__name__})")
    eng: Optional[StepEngine] = getattr(model, _ENGINE_ATTR, None)
    if eng is not None and eng.S_max >= need_seq and eng.max_T >= need_T:
        return eng
    dev = model.lm_head.weight.device
    if dev.type != "cuda":
        raise cabi.LadeHipError("USE_LADE=1 needs the model on a GPU (the HIP hot path has no CPU fallback)")
    dtype = model.lm_head.weight.dtype
    eng = StepEngine(config_from_hf(model), weights_from_hf(model), dtype=dtype, device=dev,
                     max_seq=max(need_seq, 2048), max_T=max(need_T, 512))
    setattr(model, _ENGINE_ATTR, eng)
    return eng


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
    key = (W, N, G, bool(CONFIG_MAP.get("POOL_FROM_PROMPT", 0)), R)
    dec = getattr(self, "_lade_decoder", None)
    if dec is None or getattr(dec, "_key", None) != key or dec.e is not eng:
        lp = None
        if R > 1:
            from .parallel import LPContext
            lp = LPContext(rank=CONFIG_MAP.get("LOCAL_RANK", 0), world=R)
        dec = LookaheadDecoder(eng, W, N, G, pool_from_prompt=key[3], lp=lp, use_graph=bool(int(os.environ.get("LADE_GRAPH", "1"))) and R == 1)
        dec._key = key
        self._lade_decoder = dec
    if do_sample:
        out = dec.sample(prompt, max_length, warp=warp, eos_token_id=eos_token_id, rng=random)
    else:
        out = dec.greedy(prompt, max_length, eos_token_id=eos_token_id, rng=random)
    if CONFIG_MAP.get("DEBUG", 0) and CONFIG_MAP.get("LOCAL_RANK", 0) == 0:
        print("\n==========================ACCELERATION===SUMMARY======================================")
        print("Generated tokens: ", out.generated, "Total steps: ", out.steps, " Compression ratio: ", round(out.generated / max(out.steps, 1), 2))
        print("======================================================================================", end="")
    if chat and getattr(self, "tokenizer", None) is not None and CONFIG_MAP.get("LOCAL_RANK", 0) == 0:
        print(self.tokenizer.decode(out.tokens[len(prompt):], skip_special_tokens=True), flush=True, end="")
    res = torch.tensor([out.tokens], dtype=input_ids.dtype, device=input_ids.device)
    if streamer is not None:
        streamer.put(res[:, len(prompt):].cpu())
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
