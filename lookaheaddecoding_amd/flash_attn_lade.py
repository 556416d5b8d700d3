"""`flash_attn_func(q, k, v, dropout_p, softmax_scale, causal, lookahead=[7 ints])` - the operator boundary of the reference.

The reference's flash path calls an out-of-tree CUDA wheel (`flash_attn_lade`, Viol2000/flash-attention-lookahead v2.3.3) at
lade/models/modeling_llama.py:705-713 with

    q [1, T, H, d], k / v [1, S, Hkv, d]          token-major ("bshd"), S = kv_cache + T, rows in the FLASH order of :1471-1485
    lookahead = [window, level, n_guess, kv_cache, fill_offset, guess_offset, 0]       built at :1184-1187, kv_cache set at :666

This module is that entry point on the MI355X kernels: a reference maintainer binds `from lookaheaddecoding_amd.flash_attn_lade import
flash_attn_func` where the reference imports the wheel (modeling_llama.py:58-63) and nothing else changes.  The 7-tuple maps onto the
closed-form mask of `lade_attn_fwd` (include/lade_hip.h: lade_mask_params) as

    s = window            gs = level - 1          lguess = n_guess * gs          P = kv_cache
    level_offset = guess_offset                   dist_offset = fill_offset - guess_offset            layout = 1 (flash row order)

and an all-zero tuple (apart from kv_cache) is the plain causal call of a prefill (the reference's own escape at :709).

K / V arrive token-major while the kernel streams K [Hkv][S_max][d] and V TRANSPOSED [Hkv][d][S_max]; `lade_kv_pack_bshd` re-lays them
(one launch, each byte read and written once).  That costs about what the reference's own per-layer `torch.cat` of the whole cache costs
(:626-629) - 37 MB read + 37 MB written at the BASELINE shape: measured 18 us on top of the 18 us attention pair (profiles/r5_flash_boundary_bench.txt) - and is why the product path (`StepEngine`) binds one level higher,
at the model step, where the cache stays resident in the kernel's layout.  This adapter exists so that the operator
the reference actually calls has a drop-in and a parity test (tests/test_gpu_flash_boundary.py).
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch

from . import cabi, ops
from .ops import StepMask

_WS: dict = {}          # (device, stream, dtype, Hkv, d) -> (k_cache, vt_cache) workspaces, grown in 256-key steps.  Keyed by the CURRENT STREAM too:
                        # calls on one stream reuse the pair in stream order (layer after layer); calls on another stream get their own pair instead
                        # of racing on this one


def mask_from_lookahead(lookahead: Optional[Sequence[int]], seqlen_q: int, seqlen_k: int) -> StepMask:
    """The reference's 7-tuple -> the kernel's closed-form mask.  Raises AssertionError exactly where the reference's own
    `_flash_attention_forward` does (lade/models/modeling_llama.py:706-709): the tuple must describe the tensors it comes with."""
    if lookahead is None or (sum(lookahead[:3]) + sum(lookahead[4:]) == 0):
        # plain causal (prefill, or a model that does not use the lookahead branch): the last seqlen_q keys are the queries' own
        return StepMask(T=seqlen_q, P=seqlen_k - seqlen_q, is_prefill=True)
    assert len(lookahead) == 7, f"lookahead must hold 7 integers, got {lookahead}"
    window, level, guess, kv_cache, fill_offset, guess_offset, _ = (int(x) for x in lookahead)
    want_q = window * (level - 1) + (level - 1) * guess + fill_offset
    want_k = kv_cache + want_q
    assert want_q == seqlen_q and want_k == seqlen_k, f"Setups: {list(lookahead)}, {want_q}, {want_k}, {seqlen_q}, {seqlen_k}"
    gs = max(level - 1, 1)
    return StepMask(T=seqlen_q, P=kv_cache, is_prefill=False, s=window, lguess=guess * gs, gs=gs, level_offset=guess_offset,
                    dist_offset=fill_offset - guess_offset, layout=1)


def lookahead_tuple(n_input: int, level_sizes: Sequence[int], n_guess: int, kv_cache: int) -> list:
    """The tuple as the reference builds it (lade/models/modeling_llama.py:1181-1187, kv_cache filled in at :666) - tests and callers that
    start from a step description rather than from the reference's own model code."""
    level_offset = n_input - 1
    dist_offset = 1 + level_sizes[0] - level_sizes[-1]
    return [level_sizes[-1], len(level_sizes) + 1, n_guess, kv_cache, level_offset + dist_offset, level_offset, 0]


def _workspace(dev: torch.device, dtype: torch.dtype, Hkv: int, d: int, S: int):
    S_max = ((S + 255) // 256) * 256
    key = (dev, cabi.stream_ptr(), dtype, Hkv, d)
    ws = _WS.get(key)
    if ws is None or ws[0].shape[1] < S_max:
        # zero-filled once: rows past S are never visible to a query, but the last tile's V^T columns are multiplied by zero
        # probabilities - they must be finite the first time they are touched (the kernel zeroes them in LDS as well)
        ws = (torch.zeros(Hkv, S_max, d, dtype=dtype, device=dev), torch.zeros(Hkv, d, S_max, dtype=dtype, device=dev))
        _WS[key] = ws
    return ws


def flash_attn_func(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, dropout_p: float = 0.0, softmax_scale: Optional[float] = None,
                    causal: bool = True, lookahead: Optional[Sequence[int]] = None, *, n_splits: Optional[int] = None) -> torch.Tensor:
    """Drop-in for `flash_attn_lade.flash_attn_func` on the lookahead path.  q [1, T, H, d]; k, v [1, S, Hkv, d] (S = kv_cache + T, the
    T new rows last, in the same flash order as q); returns [1, T, H, d] in q's dtype.  `n_splits` is this library's own launch
    parameter (None = its rule)."""
    if not (q.is_cuda and k.is_cuda and v.is_cuda):
        raise cabi.LadeHipError("flash_attn_func needs GPU tensors (the HIP path has no CPU fallback)")
    if q.dim() != 4 or k.dim() != 4 or v.shape != k.shape:
        raise cabi.LadeHipError(f"flash_attn_func: q {tuple(q.shape)} k {tuple(k.shape)} v {tuple(v.shape)} (expected [1, T, H, d] / [1, S, Hkv, d])")
    B, T, H, d = q.shape
    _, S, Hkv, dk = k.shape
    if B != 1 or k.shape[0] != 1:
        raise cabi.LadeHipError("flash_attn_func: single batch only (the reference asserts it, lade/models/modeling_llama.py:1448)")
    if dk != d or H % Hkv != 0 or q.dtype != k.dtype or q.dtype != v.dtype:
        raise cabi.LadeHipError(f"flash_attn_func: H={H} Hkv={Hkv} d={d}/{dk} dtypes {q.dtype} {k.dtype} {v.dtype}")
    if dropout_p != 0.0:
        raise cabi.LadeHipError("flash_attn_func: inference only (dropout_p must be 0; the reference passes 0.0 outside training, :640)")
    if not causal:
        raise cabi.LadeHipError("flash_attn_func: the lookahead kernel is causal (the reference always passes causal=True, :712)")
    mask = mask_from_lookahead(lookahead, T, S)
    kc, vt = _workspace(q.device, q.dtype, Hkv, d, S)
    S_max = kc.shape[1]
    kk, vv = k[0], v[0]
    esz = kk.element_size()
    if not (kk.stride(2) == 1 and vv.stride() == kk.stride() and (kk.stride(0) * esz) % 16 == 0 and (kk.stride(1) * esz) % 16 == 0):
        kk, vv = kk.contiguous(), vv.contiguous()          # anything but a strided view with dense head rows: one plain copy first
    with torch.cuda.device(q.device):
        # the reference passes `key_states.transpose(1, 2)`: a VIEW of [Hkv][S][d] - read through its strides, no copy
        cabi.call("lade_kv_pack_bshd", cabi.ptr(kk), cabi.ptr(vv), kk.stride(0), kk.stride(1), cabi.ptr(kc), cabi.ptr(vt), S, Hkv, d, S_max, esz)
        q2 = q[0].reshape(T, H * d)
        if not q2.is_contiguous():
            q2 = q2.contiguous()
        out = ops.attn_fwd(q2, kc, vt, mask, H=H, Hkv=Hkv, d=d, n_splits=n_splits,
                           scale=softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(d))
    return out.view(1, T, H, d)
