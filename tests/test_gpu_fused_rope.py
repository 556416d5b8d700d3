"""RoPE + KV append fused into the attention launch (lade_attn_args.n_parts > 0) and the work-group shape as a launch parameter
(lade_attn_args.wg_rows): q, the cache rows and the attention output must equal the two-launch form (lade_rope_kv_append_parts +
lade_attn_fwd: lade/models/modeling_llama.py:321-346, :510-516, :520-541) BIT FOR BIT - same partial sums in the same order, same
per-op rounding, same tiles in the same order - in every shape the step engine can launch: MHA and GQA, one and several row blocks per
KV head, new rows inside one split or spilling over two, short caches (the path that requests nothing before the rows are stored),
1..4 partials, explicit positions and per-step gathered rows, device-side cache length, bf16 and f16, d = 128 and 64."""
import math
import random

import pytest
import torch

pytestmark = pytest.mark.gpu

import lade_oracle as O


def _experimental():
    from lookaheaddecoding_amd import cabi
    return cabi.experimental()


# the fused forms are bit-identical but SLOWER at every BASELINE shape: since round 6 they are compiled only by
# `make -C lookaheaddecoding_amd/csrc EXPERIMENTAL=1`; the default library refuses n_parts > 0 (test_default_build_refuses_the_fused_forms)
needs_experimental = pytest.mark.skipif("not __import__('lookaheaddecoding_amd.cabi', fromlist=['x']).experimental()",
                                        reason="library built without EXPERIMENTAL=1: the fused RoPE + KV append forms are not in it")


def _tables(d, n, dtype):
    from lookaheaddecoding_amd.engine import rope_tables
    return rope_tables(d, n, 10000.0, dtype, "cuda")


def _case(H, Hkv, d, T, P, n_parts, n_splits, wg, dtype, mask, seed, use_pos=True, dyn=False, S_max=None):
    from lookaheaddecoding_amd import ops
    g = torch.Generator(device="cuda").manual_seed(seed)
    row_w = (H + 2 * Hkv) * d
    S_max = S_max or ((P + T + 63) // 64 * 64 + 64)
    parts = torch.randn(n_parts, T, row_w, device="cuda", generator=g) * (1.0 / math.sqrt(n_parts))
    kc = torch.randn(Hkv, S_max, d, device="cuda", generator=g).to(dtype)
    vt = torch.randn(Hkv, d, S_max, device="cuda", generator=g).to(dtype)
    kc[:, P:] = float("nan")                       # rows the step must write (and rows past P+T it must never read)
    vt[:, :, P:] = float("nan")
    cos, sin = _tables(d, 4096, dtype)
    rnd = random.Random(seed)
    pos = torch.tensor([rnd.randrange(4000) for _ in range(T)], dtype=torch.int32, device="cuda")
    dyn_P = torch.tensor([P] + [0] * 63, dtype=torch.int32, device="cuda") if dyn else None
    m = mask if not dyn else ops.StepMask(**{**mask.__dict__, "P": 0})
    if use_pos:
        rpos, rcos, rsin = pos, cos, sin
    else:                                          # per-step gathered rows, positions = None (table row t)
        rpos, rcos, rsin = None, cos[pos.long()].contiguous(), sin[pos.long()].contiguous()
    # ---- two launches
    k0, v0 = kc.clone(), vt.clone()
    q0 = torch.empty(T, H * d, dtype=dtype, device="cuda")
    iota = torch.arange(T, dtype=torch.int32, device="cuda")
    ops.rope_kv_append_parts(parts, n_parts, q0, rpos if rpos is not None else iota, rcos, rsin, k0, v0, T, P, H=H, Hkv=Hkv, d=d, dyn_P=dyn_P)
    # (3 or 4 partials do not fit the 128-row shape's registers: the fused launch then runs 64-row blocks, attn.hip launch_fwd_npc)
    wg_ref = 64 if (n_parts > 2 and wg in (0, 128)) else wg
    out0 = ops.attn_fwd(q0, k0, v0, m, H=H, Hkv=Hkv, d=d, n_splits=n_splits, dyn_P=dyn_P, wg_rows=wg_ref)
    # ---- one launch
    k1, v1 = kc.clone(), vt.clone()
    out1 = ops.attn_fwd(None, k1, v1, m, H=H, Hkv=Hkv, d=d, n_splits=n_splits, dyn_P=dyn_P, wg_rows=wg, qkv_parts=parts, n_parts=n_parts,
                        positions=rpos, cos=rcos, sin=rsin)
    torch.cuda.synchronize()
    tag = (H, Hkv, d, T, P, n_parts, n_splits, wg, dtype, use_pos, dyn)

    def same(k1, v1, out1, what):
        assert torch.equal(k1[:, :P + T].view(torch.int16), k0[:, :P + T].view(torch.int16)), ("K rows", what, tag)
        assert torch.equal(v1[:, :, :P + T].view(torch.int16), v0[:, :, :P + T].view(torch.int16)), ("V^T rows", what, tag)
        assert bool(torch.isnan(k1[:, P + T:].float()).all()) and bool(torch.isnan(v1[:, :, P + T:].float()).all()), ("rows past P+T touched", what, tag)
        assert torch.isfinite(out1.float()).all(), (what, tag)
        assert torch.equal(out1.view(torch.int16), out0.view(torch.int16)), ("attention output", what, tag, (out1.float() - out0.float()).abs().max().item())

    same(k1, v1, out1, "every split rebuilds q")
    if n_splits > 1:
        # ---- one launch, producer mode: dedicated work-groups do the RoPE + append once per KV head and hand it over inside the launch; three
        # launches in a row on ONE flag buffer (lade_attn_combine leaves it zero again), the q buffer poisoned before each
        flags = torch.zeros(64, dtype=torch.int32, device="cuda")
        for rep in range(3):
            k2, v2 = kc.clone(), vt.clone()
            q2 = torch.full((T, H * d), float("nan"), dtype=dtype, device="cuda")
            out2 = ops.attn_fwd(q2, k2, v2, m, H=H, Hkv=Hkv, d=d, n_splits=n_splits, dyn_P=dyn_P, wg_rows=wg, qkv_parts=parts, n_parts=n_parts,
                                positions=rpos, cos=rcos, sin=rsin, sync_flags=flags)
            torch.cuda.synchronize()
            assert torch.equal(q2.view(torch.int16), q0.view(torch.int16)), ("q rows", "producer mode", rep, tag)
            same(k2, v2, out2, f"producer mode, launch {rep}")
            assert int(flags.abs().sum().item()) == 0, ("flags not reset", rep, tag)
    return out1


def _lookahead_mask(W, N, g, P):
    from lookaheaddecoding_amd import ops
    ls = [W - 1] + [W] * (N - 2)
    return ops.StepMask.from_levels(1, ls, g * (N - 1), N - 1, P)


@needs_experimental
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_rope_equals_two_launches_mha(dtype):
    """Llama-2-7B head shape, the bench's own steps: T = 60 / 64 / 120, 2 partials, the split counts and shapes the tuner may pick"""
    for (g, P, ns, wg, npart) in ((0, 2076, 6, 128, 2), (0, 2076, 8, 64, 2), (1, 2219, 6, 128, 2), (15, 2076, 6, 128, 2), (15, 1000, 4, 64, 3),
                                  (0, 2076, 6, 32, 1), (4, 517, 3, 128, 4), (0, 2076, 1, 128, 2)):
        m = _lookahead_mask(15, 5, g, P)
        _case(32, 32, 128, m.T, P, npart, ns, wg, dtype, m, seed=P + g + ns)


@needs_experimental
def test_fused_rope_equals_two_launches_gqa_and_row_blocks():
    """Llama-2-70B group shape (8 heads per KV head: 480 rows = 4 row blocks at 128 rows, 8 at 64, 15 at 32 share one K / V stream and
    write the same new rows) and a 240-row MHA step (config 4 with candidates: 2 row blocks; the new rows span 4-5 tiles and two splits)"""
    for (H, Hkv, W, N, g, P, ns, wg, npart) in ((16, 2, 15, 5, 0, 2076, 6, 128, 2), (16, 2, 15, 5, 0, 2076, 3, 64, 2), (16, 2, 15, 5, 2, 700, 2, 32, 2),
                                                (8, 8, 20, 7, 20, 1000, 5, 128, 2), (8, 8, 20, 7, 20, 1030, 5, 64, 3), (8, 8, 20, 7, 3, 1030, 8, 128, 2)):
        m = _lookahead_mask(W, N, g, P)
        _case(H, Hkv, 128, m.T, P, npart, ns, wg, torch.bfloat16, m, seed=H + P + ns)


@needs_experimental
def test_fused_rope_short_caches_and_the_deferred_request_path():
    """caches so short that a split reaches the new rows within its first ring stages: nothing may be requested before the rows are stored"""
    from lookaheaddecoding_amd import ops
    for (T, P, ns, wg) in ((60, 0, 1, 128), (60, 3, 1, 64), (60, 70, 2, 128), (17, 130, 2, 32), (120, 64, 1, 128), (120, 200, 3, 64), (9, 190, 4, 32),
                           (60, 128, 2, 128), (64, 192, 4, 128)):
        m = ops.StepMask(T=T, P=P, is_prefill=True)
        _case(4, 4, 128, T, P, 2, ns, wg, torch.bfloat16, m, seed=T + P)
        _case(4, 2, 64, T, P, 2, ns, wg, torch.float16, m, seed=T + P + 1)


@needs_experimental
def test_fused_rope_prefill_chunks_gathered_rows_and_device_cache_length():
    """causal chunks (the tiles behind a row block's last token are neither requested nor produced by that block), cos / sin rows gathered
    per step (positions = None), cache length read from the device"""
    from lookaheaddecoding_amd import ops
    for (T, P, ns, wg) in ((200, 512, 3, 128), (256, 256, 2, 64), (130, 1000, 5, 128)):
        m = ops.StepMask(T=T, P=P, is_prefill=True)
        _case(4, 4, 128, T, P, 2, ns, wg, torch.bfloat16, m, seed=T, use_pos=False)
    m = _lookahead_mask(15, 5, 4, 1500)
    _case(8, 8, 128, m.T, 1500, 2, 5, 128, torch.bfloat16, m, seed=3, dyn=True)
    _case(8, 2, 128, m.T, 1500, 2, 2, 64, torch.bfloat16, m, seed=4, dyn=True, use_pos=False)
    m = _lookahead_mask(5, 3, 3, 40)
    _case(4, 4, 64, m.T, 40, 2, 1, 32, torch.bfloat16, m, seed=5)


def test_work_group_shape_is_a_launch_parameter_with_equal_results_up_to_rounding():
    """wg_rows 128 / 64 / 32: same attention within the attention tolerance (the shapes merge their key parts in different orders), each
    against the dense fp32 oracle"""
    from lookaheaddecoding_amd import ops
    torch.manual_seed(9)
    H, Hkv, d, P = 8, 2, 128, 777
    ls, lguess, gs = [14, 15, 15, 15], 20, 4
    T = 1 + sum(ls) + lguess
    S = P + T
    lay = O.StepLayout(ids=[0] * T, positions=[], n_input=1, level_sizes=ls, lguess=lguess, is_prefill=False, window=15)
    vis = O.dense_mask(lay, P, gs)
    q = torch.randn(T, H, d).bfloat16()
    k = torch.randn(Hkv, S + 64, d).bfloat16()
    v = torch.randn(Hkv, S + 64, d).bfloat16()
    ref = O.attention_dense(q.float().transpose(0, 1), k.float()[:, :S], v.float()[:, :S], vis).transpose(0, 1).reshape(T, H * d)
    S_max = (S + 127) // 64 * 64
    kc = torch.zeros(Hkv, S_max, d, dtype=torch.bfloat16); kc[:, :S + 64] = k
    vt = torch.zeros(Hkv, d, S_max, dtype=torch.bfloat16); vt[:, :, :S + 64] = v.transpose(1, 2)
    m = ops.StepMask.from_levels(1, ls, lguess, gs, P)
    for wg in (0, 128, 64, 32):
        for ns in (1, 2, 5):
            out = ops.attn_fwd(q.reshape(T, H * d).cuda(), kc.cuda(), vt.cuda(), m, H=H, Hkv=Hkv, d=d, n_splits=ns, wg_rows=wg).float().cpu()
            assert torch.allclose(out, ref, atol=2e-2, rtol=2e-2), (wg, ns, (out - ref).abs().max().item())


@needs_experimental
def test_fused_rope_argument_validation():
    from lookaheaddecoding_amd import cabi, ops
    m = ops.StepMask(T=8, P=64, is_prefill=True)
    kc = torch.zeros(2, 128, 128, dtype=torch.bfloat16, device="cuda")
    vt = torch.zeros(2, 128, 128, dtype=torch.bfloat16, device="cuda")
    cos, sin = _tables(128, 256, torch.bfloat16)
    parts = torch.zeros(5, 8, 6 * 128, device="cuda")
    with pytest.raises(cabi.LadeHipError, match="1..4"):
        ops.attn_fwd(None, kc, vt, m, H=2, Hkv=2, d=128, n_splits=1, qkv_parts=parts, n_parts=5, positions=None, cos=cos, sin=sin)
    with pytest.raises(cabi.LadeHipError, match="halves"):
        bad = cos.clone(); bad[:, 64:] += 1
        ops.attn_fwd(None, kc, vt, m, H=2, Hkv=2, d=128, n_splits=1, qkv_parts=parts, n_parts=2, positions=None, cos=bad, sin=sin)
    with pytest.raises(cabi.LadeHipError, match="wg_rows"):
        ops.attn_fwd(torch.zeros(8, 256, dtype=torch.bfloat16, device="cuda"), kc, vt, m, H=2, Hkv=2, d=128, n_splits=1, wg_rows=48)


@needs_experimental
def test_engine_step_with_and_without_the_fused_launch_bit_identical(monkeypatch):
    """whole bf16 decoding runs at the 7B width (2 layers, attention and MLP live): token ids, step counts and the K / V rows with RoPE
    fused into the attention launch - both forms - == with the RoPE launch of its own, eager and hipGraph.  ONE engine, the form switched
    between runs (the in-step tuner is off, so the GEMM table is the same throughout and only the attention launch differs).  Every form
    runs 64-row work-groups: a fused launch fed by 3 or 4 partials cannot take the 128-row shape (registers, attn.hip launch_fwd_npc), and the
    isolated GEMM tuner does sometimes pick 4 splits for the 128-row qkv class - the 128- and 64-row shapes merge their key parts in a
    different order and differ by one bf16 spacing in a handful of rows (seen as 5 prompt rows of a layer's K / V, in the runs where the tuner
    had picked 4 splits: tools/fused_rope_diag.py)."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    from lookaheaddecoding_amd.weights import make_config, random_weights_torch
    monkeypatch.setenv("LADE_TUNE_STEP", "0")
    cfg = make_config("llama2-7b", layers=2)
    w = random_weights_torch(cfg, seed=2, dtype=torch.bfloat16, device="cuda", std=0.03)
    rng = random.Random(5)
    prompt = [rng.randrange(3, cfg["vocab"]) for _ in range(150)]
    eng = StepEngine(cfg, w, dtype=torch.bfloat16, device="cuda", max_seq=512, max_T=128)
    LookaheadDecoder(eng, 15, 5, 15).greedy(prompt, len(prompt) + 8, rng=random.Random(1))          # throw-away
    outs = {}
    for fuse in (0, 1, 2, 0):                                 # two launches | every split rebuilds q | producer work-groups + in-launch hand-off | again
        eng.attn_default = (fuse, 64, 0)
        assert not eng.attn_cfg                               # (no tuned entry overrides the default)
        for graph in (False, True):
            dec = LookaheadDecoder(eng, 15, 5, 15, use_graph=graph)
            o = dec.greedy(prompt, len(prompt) + 24, rng=random.Random(1))
            n_keep = len(o.tokens) - 1                        # rows of accepted tokens: identical whatever speculative rows lie behind them
            outs[(fuse, graph, len([k for k in outs if k[:2] == (fuse, graph)]))] = (o.tokens, o.steps, eng.kv.view(eng.L, 2, -1).clone(), n_keep)
        qkv = eng.gemm_cfg[("wqkv", 64)]
        assert qkv is not None and qkv[2] <= 4                # the fused forms really ran (a split-K qkv GEMM with <= 4 partials)
    ref = outs[(0, False, 0)]
    Hkv, d, S = cfg["kv_heads"], cfg["head_dim"], 512
    bad = []
    for key, (tok, steps, kv, n_keep) in outs.items():
        assert tok == ref[0] and steps == ref[1], key
        a = outs[(0, key[1], 0)]
        ka, kb = a[2][:, 0].view(-1, Hkv, S, d)[:, :, :a[3]], kv[:, 0].view(-1, Hkv, S, d)[:, :, :a[3]]
        va, vb = a[2][:, 1].view(-1, Hkv, d, S)[:, :, :, :a[3]], kv[:, 1].view(-1, Hkv, d, S)[:, :, :, :a[3]]
        dk = (ka.view(torch.int16) != kb.view(torch.int16)).nonzero()
        dv = (va.contiguous().view(torch.int16) != vb.contiguous().view(torch.int16)).nonzero()
        if len(dk) or len(dv):
            bad.append((key, "K rows", sorted(set(dk[:, 2].tolist()))[:10], "layers", sorted(set(dk[:, 0].tolist())), "V columns", sorted(set(dv[:, 3].tolist()))[:10],
                        "layers", sorted(set(dv[:, 0].tolist())), "n_keep", a[3], n_keep))
    assert not bad, bad


def test_default_build_refuses_the_fused_forms():
    """a library built without EXPERIMENTAL=1 answers n_parts > 0 with an error code, never with a silent fallback"""
    from lookaheaddecoding_amd import cabi, ops
    if cabi.experimental():
        pytest.skip("experimental build: the fused forms are present")
    cos, sin = _tables(128, 64, torch.bfloat16)
    kc = torch.zeros(2, 64, 128, device="cuda", dtype=torch.bfloat16)
    vt = torch.zeros(2, 128, 64, device="cuda", dtype=torch.bfloat16)
    parts = torch.zeros(2, 4, 6 * 128, device="cuda")
    m = ops.StepMask(T=4, P=8, is_prefill=True)
    with pytest.raises(cabi.LadeHipError, match="not in this build"):
        ops.attn_fwd(None, kc, vt, m, H=2, Hkv=2, d=128, n_splits=1, qkv_parts=parts, n_parts=2, positions=None, cos=cos, sin=sin)
