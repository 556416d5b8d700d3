"""Synthetic Llama-shaped weights (no hub, no tokenizer) and the shape presets of BASELINE.json.

The reference loads real checkpoints through HF (`lade/utils.py:89-101`); there is no network
here, so every test / bench builds random-weight models of the named architecture.  Weights are
plain tensors in nn.Linear layout ([out_features, in_features]) keyed

    embed [V,hid]  norm [hid]  lm_head [V,hid]
    layers.{i}.ln1 / ln2 [hid]   layers.{i}.wq [H*d,hid]  wk, wv [Hkv*d,hid]  wo [hid,H*d]
    layers.{i}.wg, wu [inter,hid]   layers.{i}.wd [hid,inter]

which is what both `StepEngine` (HIP) and the CPU oracle consume.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

# shapes: SURVEY.md section 8 (C1..C5) + tiny shapes for parity tests
PRESETS: Dict[str, dict] = {
    # TinyLlama-1.1B shape (BASELINE config 1)
    "tinyllama-1.1b": dict(hidden=2048, inter=5632, layers=22, heads=32, kv_heads=4, head_dim=64, vocab=32000),
    # Llama-2-7B shape (BASELINE configs 2/3)
    "llama2-7b": dict(hidden=4096, inter=11008, layers=32, heads=32, kv_heads=32, head_dim=128, vocab=32000),
    # CodeLlama-13B shape (BASELINE config 4)
    "codellama-13b": dict(hidden=5120, inter=13824, layers=40, heads=40, kv_heads=40, head_dim=128, vocab=32016),
    # Llama-2-70B shape (BASELINE config 5)
    "llama2-70b": dict(hidden=8192, inter=28672, layers=80, heads=64, kv_heads=8, head_dim=128, vocab=32000),
    # parity-test shapes
    "tiny-d64": dict(hidden=128, inter=256, layers=2, heads=2, kv_heads=1, head_dim=64, vocab=256),
    "tiny-d128": dict(hidden=256, inter=384, layers=2, heads=2, kv_heads=2, head_dim=128, vocab=320),
    "tiny-d16": dict(hidden=64, inter=176, layers=2, heads=4, kv_heads=2, head_dim=16, vocab=128),
}


def make_config(name_or_cfg, **overrides) -> dict:
    cfg = dict(PRESETS[name_or_cfg]) if isinstance(name_or_cfg, str) else dict(name_or_cfg)
    cfg.setdefault("eps", 1e-5 if cfg.get("hidden", 0) >= 2048 else 1e-6)
    cfg.setdefault("rope_theta", 10000.0)
    cfg.setdefault("max_pos", 4096)
    cfg.update(overrides)
    assert cfg["heads"] % cfg["kv_heads"] == 0
    return cfg


def weight_shapes(cfg: dict) -> Dict[str, tuple]:
    hid, inter, H, Hkv, d, V = cfg["hidden"], cfg["inter"], cfg["heads"], cfg["kv_heads"], cfg["head_dim"], cfg["vocab"]
    shapes = {"embed": (V, hid), "norm": (hid,), "lm_head": (V, hid)}
    for i in range(cfg["layers"]):
        p = f"layers.{i}."
        shapes.update({p + "ln1": (hid,), p + "ln2": (hid,), p + "wq": (H * d, hid), p + "wk": (Hkv * d, hid),
                       p + "wv": (Hkv * d, hid), p + "wo": (hid, H * d), p + "wg": (inter, hid),
                       p + "wu": (inter, hid), p + "wd": (hid, inter)})
    return shapes


def random_weights_numpy(cfg: dict, seed: int = 0, std: float = 0.02, tie_lm_head: bool = False) -> Dict[str, np.ndarray]:
    """Deterministic across machines (numpy RandomState stream, fixed key order).  HF init:
    normal(0, std) for linears/embeddings, ones for RMSNorm weights
    (lade/models/modeling_llama.py:934-943).  Meant for small (test) shapes."""
    rs = np.random.RandomState(seed)
    out: Dict[str, np.ndarray] = {}
    for k, shp in weight_shapes(cfg).items():
        if len(shp) == 1:
            out[k] = np.ones(shp, dtype=np.float32)
        else:
            out[k] = (rs.standard_normal(shp) * std).astype(np.float32)
    if tie_lm_head:
        out["lm_head"] = out["embed"].copy()
    return out


def random_weights_torch(cfg: dict, seed: int = 0, std: float = 0.02, dtype=torch.bfloat16, device="cuda",
                         tie_lm_head: bool = False) -> Dict[str, torch.Tensor]:
    """Large shapes: generate directly on the target device in `dtype` (7B in bf16 = 13.5 GB)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out: Dict[str, torch.Tensor] = {}
    for k, shp in weight_shapes(cfg).items():
        if len(shp) == 1:
            out[k] = torch.ones(shp, dtype=dtype, device=device)
        else:
            w = torch.empty(shp, dtype=dtype, device=device)
            w.normal_(0.0, std, generator=g)
            out[k] = w
    if tie_lm_head:
        out["lm_head"] = out["embed"]
    return out


def to_torch(weights: Dict[str, np.ndarray], dtype=torch.float32, device="cpu") -> Dict[str, torch.Tensor]:
    return {k: torch.as_tensor(v).to(device=device, dtype=dtype) for k, v in weights.items()}
