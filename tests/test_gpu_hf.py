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
