"""CPU: the drop-in surface of the reference (`lade.*`, lade/utils.py, lade/__init__.py, lade/lade_distributed.py)."""
import os

import pytest


def test_lade_alias_exports_reference_names():
    import lade
    import lookaheaddecoding_amd as impl
    for name in ("augment_all", "augment_llama", "augment_generate", "config_lade", "log_history", "save_log", "get_device", "distributed"):
        assert getattr(lade, name) is getattr(impl, name)
    import lade.decoding as D
    assert D.CONFIG_MAP is impl.decoding.CONFIG_MAP and isinstance(D.FUNC_MAP, dict)


def test_config_lade_semantics(capsys):
    import lade
    lade.decoding.CONFIG_MAP.clear()
    lade.config_lade(LEVEL=5, WINDOW_SIZE=15, GUESS_SET_SIZE=15, DEBUG=1, POOL_FROM_PROMPT=True, USE_FLASH=0)
    cm = lade.decoding.CONFIG_MAP
    assert (cm["LEVEL"], cm["WINDOW_SIZE"], cm["GUESS_SET_SIZE"], cm["DEBUG"], cm["POOL_FROM_PROMPT"]) == (5, 15, 15, 1, True)
    assert cm["log"] == [] and not lade.distributed() and lade.get_device() == 0
    cm["log"].append([10, 5, 2.0]); cm["log"].append([6, 3, 2.0])
    lade.log_history()
    assert "OVERALL GEN:  16  STEPS:  8  AVG COMPRESS RATIO:  2.0" in capsys.readouterr().out
    lade.config_lade(LEVEL=4)                      # every call resets the log (lade/utils.py:37)
    assert cm["log"] == [] and cm["LEVEL"] == 4 and cm["WINDOW_SIZE"] == 15
    lade.decoding.CONFIG_MAP.clear()


def test_save_log(tmp_path):
    import torch
    import lade
    lade.config_lade(DEBUG=1)
    lade.decoding.CONFIG_MAP["log"].append([4, 2, 2.0])
    p = tmp_path / "log.pt"
    lade.save_log(str(p))
    assert torch.load(str(p)) == [[4, 2, 2.0]]
    lade.decoding.CONFIG_MAP.clear()


def test_augment_generate_patches_and_env_gate(monkeypatch):
    """USE_LADE unset -> the saved HF function runs (lade/decoding.py:15-34)."""
    import lade
    from transformers import GenerationMixin
    from lookaheaddecoding_amd import hf
    orig = GenerationMixin._sample
    try:
        lade.augment_generate()
        assert GenerationMixin._sample is hf._sample_proxy
        called = {}
        monkeypatch.setitem(lade.decoding.FUNC_MAP, "_sample", lambda self, *a, **k: called.setdefault("orig", (a, k)) or "ORIG")
        monkeypatch.delenv("USE_LADE", raising=False)
        assert GenerationMixin._sample(object(), "ids", logits_processor=[], stopping_criteria=[], generation_config=None) is not None
        assert "orig" in called
    finally:
        GenerationMixin._sample = orig
        lade.decoding.FUNC_MAP.pop("_sample", None)


def test_use_lade_without_gpu_fails_loudly(monkeypatch):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import lade
    from transformers import LlamaConfig, LlamaForCausalLM
    from lookaheaddecoding_amd import cabi, hf
    cfg = LlamaConfig(vocab_size=64, hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=1, num_key_value_heads=1,
                      max_position_embeddings=128)
    model = LlamaForCausalLM(cfg).eval()
    lade.config_lade(LEVEL=3, WINDOW_SIZE=3, GUESS_SET_SIZE=3)
    with pytest.raises(cabi.LadeHipError):
        hf.jacobi_greedy_search_multilevel(model, torch.tensor([[1, 2, 3]]), max_length=8)
    lade.decoding.CONFIG_MAP.clear()


def test_jforward_multilevel_has_the_reference_signature():
    """The model-step boundary keeps the reference's parameter names and order (lade/models/modeling_llama.py:1381-1405)."""
    import inspect
    from lookaheaddecoding_amd import hf
    names = list(inspect.signature(hf.jforward_multilevel).parameters)
    assert names == ["self", "input_ids", "past_tokens", "guess_tokens", "guess_size", "not_seq", "continue_all", "level", "fill_level", "WINDOWS_SIZE",
                     "dist_workers", "local_rank", "la_mask_offset", "use_flash", "attention_mask", "position_ids", "past_key_values", "inputs_embeds",
                     "labels", "use_cache", "output_attentions", "output_hidden_states", "return_dict"]
    out = hf.StepOutput()
    for field in ("out_logits", "inp_logits", "guess_logits", "past_key_values", "kvcache_len", "step_len"):
        assert hasattr(out, field)
