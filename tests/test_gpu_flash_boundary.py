"""The operator boundary the reference actually calls - `flash_attn_func(q [1,T,H,d], k / v [1,S,Hkv,d], dropout, softmax_scale, causal=True,
lookahead=[window, level, n_guess, kv_cache, fill_offset, guess_offset, 0])`, lade/models/modeling_llama.py:705-713, tuple built at
:1184-1187, kv_cache set at :666 - driven with the tuples of EVERY step of the reference's own greedy runs (tests/golden/e2e_greedy.json,
e2e_greedy_wide.json) and compared with the oracle's eager attention (the dense mask of :115-207) after un-permuting the flash row
order of :1471-1485.  Tolerances: fp32 2e-5 / 1e-4, f16 4e-3 / 1e-2, bf16 2e-2 / 2e-2 (the attention tolerances of DESIGN 5)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import lade_oracle as O
from conftest import GOLDEN

TOL = {torch.bfloat16: dict(atol=2e-2, rtol=2e-2), torch.float16: dict(atol=4e-3, rtol=1e-2), torch.float32: dict(atol=2e-5, rtol=1e-4)}


def _steps():
    """(run tag, N, trace entry) of every step of the reference's greedy runs whose levels >= 1 have equal lengths - the steps the
    reference can feed its flash kernel at all (np.array(all_past[1:]).transpose() at :1483 needs a rectangular array)"""
    out = []
    for name in ("e2e_greedy.json", "e2e_greedy_wide.json"):
        with open(os.path.join(GOLDEN, name)) as f:
            d = json.load(f)
        for ri, run in enumerate(d["runs"]):
            for si, tr in enumerate(run["trace"]):
                ls = tr["level_sizes"]
                if not tr["is_prefill"] and len(ls) > 1 and any(x != ls[1] for x in ls[1:]):
                    continue
                out.append((f"{name}:{ri}:{si}", run["N"], tr))
    return out


def _one_step(tag, N, tr, dtype, H, Hkv, dh, rng, transposed_view, zero_tuple=False):
    from lookaheaddecoding_amd.flash_attn_lade import flash_attn_func, lookahead_tuple
    ls, lguess, n_input, P = tr["level_sizes"], tr["lguess"], tr["n_input"], tr["P"]
    gs = N - 1
    T = n_input + sum(ls) + lguess
    S = P + T
    lay = O.StepLayout(ids=[0] * T, positions=[], n_input=n_input, level_sizes=ls, lguess=lguess, is_prefill=tr["is_prefill"], window=ls[-1])
    vis = O.dense_mask(lay, P, gs)
    q = torch.randn(T, H, dh, generator=rng).to(dtype)
    k = torch.randn(S, Hkv, dh, generator=rng).to(dtype)
    v = torch.randn(S, Hkv, dh, generator=rng).to(dtype)
    ref = O.attention_dense(q.float().transpose(0, 1), k.float().transpose(0, 1), v.float().transpose(0, 1), vis).transpose(0, 1)     # [T, H, dh], eager order
    # a prefill step (prompt + L0, :1441 leaves its rows un-swapped) reaches the kernel with the tuple of :1187 too - [len(L0), 2, 0, P,
    # n_input, n_input - 1, 0], whose closed form IS the plain causal mask - and a model outside the lookahead loop with all zeros
    perm = O.flash_row_order(n_input, ls, lguess)
    tup = lookahead_tuple(n_input, ls, lguess // gs, P)
    if tr["is_prefill"] and zero_tuple:
        tup = [0, 0, 0, P, 0, 0, 0]
    rows = list(range(P)) + [P + p for p in perm]
    qf, kf, vf = q[perm], k[rows], v[rows]
    if transposed_view:            # the reference hands over `key_states.transpose(1, 2)` of a [1, Hkv, S, d] tensor: a view, not a copy
        kd = kf.transpose(0, 1).contiguous().cuda().transpose(0, 1)[None]
        vd = vf.transpose(0, 1).contiguous().cuda().transpose(0, 1)[None]
    else:
        kd, vd = kf.cuda()[None], vf.cuda()[None]
    out = flash_attn_func(qf.cuda()[None], kd, vd, 0.0, softmax_scale=None, causal=True, lookahead=tup)
    assert out.shape == (1, T, H, dh) and out.dtype == dtype
    got = out[0].float().cpu()
    assert torch.isfinite(got).all(), tag
    assert torch.allclose(got, ref[perm], **TOL[dtype]), (tag, dtype, tup, (got - ref[perm]).abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_flash_attn_func_on_every_golden_greedy_step(dtype):
    steps = _steps()
    assert len(steps) > 250
    rng = torch.Generator().manual_seed(7)
    shapes = ((4, 4, 64), (8, 2, 128)) if dtype != torch.float32 else ((4, 2, 64),)
    n = 0
    for i, (tag, N, tr) in enumerate(steps):
        if dtype != torch.float32 and i % 2 and not tr["lguess"]:
            continue                                       # 16-bit: every step with candidates, every other one without
        H, Hkv, dh = shapes[i % len(shapes)]
        _one_step(tag, N, tr, dtype, H, Hkv, dh, rng, transposed_view=bool(i % 3 == 0), zero_tuple=bool(i % 2))
        n += 1
    assert n > 120


def test_flash_attn_func_at_the_baseline_shapes():
    """config 2 (W=15 N=5 G=15, 7B heads) and config 4 (W=20 N=7 G=20, 13B heads) at a long cache, bf16 and f16, split-KV + merge"""
    from lookaheaddecoding_amd.flash_attn_lade import flash_attn_func, lookahead_tuple
    rng = torch.Generator().manual_seed(11)
    for (W, N, g, H, Hkv, dh, P) in ((15, 5, 15, 32, 32, 128, 2076), (15, 5, 0, 32, 32, 128, 2076), (20, 7, 20, 40, 40, 128, 1030), (15, 5, 15, 64, 8, 128, 2076)):
        gs = N - 1
        ls = [W - 1] + [W] * (N - 2)
        lguess = g * gs
        T = 1 + sum(ls) + lguess
        S = P + T
        lay = O.StepLayout(ids=[0] * T, positions=[], n_input=1, level_sizes=ls, lguess=lguess, is_prefill=False, window=W)
        vis = O.dense_mask(lay, P, gs)
        perm = O.flash_row_order(1, ls, lguess)
        rows = list(range(P)) + [P + p for p in perm]
        for dtype in (torch.bfloat16, torch.float16):
            q = torch.randn(T, H, dh, generator=rng).to(dtype)
            k = torch.randn(S, Hkv, dh, generator=rng).to(dtype)
            v = torch.randn(S, Hkv, dh, generator=rng).to(dtype)
            ref = O.attention_dense(q.float().transpose(0, 1), k.float().transpose(0, 1), v.float().transpose(0, 1), vis).transpose(0, 1)
            out = flash_attn_func(q[perm].cuda()[None], k[rows].cuda()[None], v[rows].cuda()[None], 0.0, causal=True,
                                  lookahead=lookahead_tuple(1, ls, g, P))[0].float().cpu()
            assert torch.allclose(out, ref[perm], **TOL[dtype]), (W, N, g, H, Hkv, dtype, (out - ref[perm]).abs().max().item())


def test_flash_attn_func_refuses_what_the_reference_refuses():
    from lookaheaddecoding_amd import cabi
    from lookaheaddecoding_amd.flash_attn_lade import flash_attn_func
    q = torch.zeros(1, 9, 2, 64, dtype=torch.bfloat16, device="cuda")
    k = torch.zeros(1, 20, 2, 64, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(AssertionError, match="Setups"):           # the reference's own seqlen identity (:706-709)
        flash_attn_func(q, k, k, 0.0, causal=True, lookahead=[2, 3, 1, 11, 2, 0, 0])
    with pytest.raises(cabi.LadeHipError):
        flash_attn_func(q, k, k, 0.1, causal=True, lookahead=[0, 0, 0, 11, 0, 0, 0])
    with pytest.raises(cabi.LadeHipError):
        flash_attn_func(q.cpu(), k.cpu(), k.cpu(), 0.0, causal=True, lookahead=[0, 0, 0, 11, 0, 0, 0])
    with pytest.raises(cabi.LadeHipError):
        flash_attn_func(torch.cat([q, q]), torch.cat([k, k]), torch.cat([k, k]), 0.0, causal=True, lookahead=[0, 0, 0, 11, 0, 0, 0])


def test_kv_pack_bshd_bit_exact():
    from lookaheaddecoding_amd import cabi
    for dtype, S, Hkv, d in ((torch.bfloat16, 197, 3, 128), (torch.float32, 70, 2, 64), (torch.float16, 64, 1, 256)):
        k = torch.randn(S, Hkv, d).to(dtype).cuda()
        v = torch.randn(S, Hkv, d).to(dtype).cuda()
        S_max = 256
        kc = torch.full((Hkv, S_max, d), 7.0, dtype=dtype, device="cuda")
        vt = torch.full((Hkv, d, S_max), 7.0, dtype=dtype, device="cuda")
        cabi.call("lade_kv_pack_bshd", cabi.ptr(k), cabi.ptr(v), k.stride(0), k.stride(1), cabi.ptr(kc), cabi.ptr(vt), S, Hkv, d, S_max, k.element_size())
        assert torch.equal(kc[:, :S], k.transpose(0, 1)) and torch.equal(vt[:, :, :S], v.permute(1, 2, 0))
        assert bool((kc[:, S:] == 7.0).all()) and bool((vt[:, :, S:] == 7.0).all())
