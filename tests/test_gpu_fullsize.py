"""BASELINE-size checks through size-independent properties (the dense oracle is too slow at these sizes):
attention at config 2 / config 5 shapes, pool at V = 32000 / G = 15, argmax at V = 32000."""
import math
import random

import pytest
import torch

pytestmark = pytest.mark.gpu


def _c2_case(T=120, P=4096, H=32, Hkv=32, d=128, seed=0, dtype=torch.bfloat16):
    from lookaheaddecoding_amd import ops
    torch.manual_seed(seed)
    W, N, G = 15, 5, 15
    gs = N - 1
    g = (T - (N - 1) * W) // gs
    S_max = (P + T + 127) // 64 * 64
    q = torch.randn(T, H * d, device="cuda").to(dtype)
    k = torch.randn(Hkv, S_max, d, device="cuda").to(dtype)
    vt = torch.randn(Hkv, d, S_max, device="cuda").to(dtype)
    mask = ops.StepMask.from_levels(1, [W - 1] + [W] * (N - 2), g * gs, gs, P)
    return q, k, vt, mask, dict(H=H, Hkv=Hkv, d=d)


def test_attention_c2_fullsize_properties():
    from lookaheaddecoding_amd import ops
    q, k, vt, mask, kw = _c2_case()
    o8 = ops.attn_fwd(q, k, vt, mask, n_splits=8, **kw).float()
    o1 = ops.attn_fwd(q, k, vt, mask, n_splits=1, **kw).float()
    o5 = ops.attn_fwd(q, k, vt, mask, n_splits=5, **kw).float()
    assert torch.isfinite(o8).all()
    # split invariance
    assert torch.allclose(o8, o1, atol=2e-2, rtol=2e-2) and torch.allclose(o5, o1, atol=2e-2, rtol=2e-2)
    # softmax rows sum to one: V = const  ->  O = const
    ones = torch.full_like(vt, 0.5)
    oc = ops.attn_fwd(q, k, ones, mask, n_splits=8, **kw).float()
    assert torch.allclose(oc, torch.full_like(oc, 0.5), atol=4e-3)
    # linearity in V
    vt2 = torch.randn_like(vt)
    oa = ops.attn_fwd(q, k, vt2, mask, n_splits=8, **kw).float()
    ob = ops.attn_fwd(q, k, (vt.float() + vt2.float()).to(vt.dtype), mask, n_splits=8, **kw).float()
    assert torch.allclose(ob, o8 + oa, atol=6e-2, rtol=3e-2)
    # rows of the cache beyond P+T are never read: poison them
    kp, vp = k.clone(), vt.clone()
    S = mask.P + mask.T
    kp[:, S:] = float("nan"); vp[:, :, S:] = float("nan")
    assert torch.equal(ops.attn_fwd(q, kp, vp, mask, n_splits=8, **kw), ops.attn_fwd(q, k, vt, mask, n_splits=8, **kw))
    # a candidate row must not depend on another candidate's keys: perturb candidate 3's K/V rows, candidate 0 rows unchanged
    T, P, gs = mask.T, mask.P, mask.gs
    c3 = P + T - mask.lguess + 3 * gs
    k2, v2 = k.clone(), vt.clone()
    k2[:, c3:c3 + gs] += 1.0; v2[:, :, c3:c3 + gs] += 1.0
    od = ops.attn_fwd(q, k2, v2, mask, n_splits=8, **kw)
    base = ops.attn_fwd(q, k, vt, mask, n_splits=8, **kw)
    r0 = T - mask.lguess
    assert torch.equal(od[r0:r0 + 3 * gs], base[r0:r0 + 3 * gs]) and torch.equal(od[:r0], base[:r0])
    assert not torch.equal(od[r0 + 3 * gs:r0 + 4 * gs], base[r0 + 3 * gs:r0 + 4 * gs])


def test_attention_gqa_70b_shape_lp_rank():
    """config 5: H=64, Hkv=8 (n_rep 8), a lookahead-parallel rank's step (T_r = 31, dist_offset > 0) vs the dense oracle."""
    import numpy as np
    import lade_oracle as O
    from lookaheaddecoding_amd import ops
    torch.manual_seed(1)
    H, Hkv, d, P = 64, 8, 128, 300
    n_input, ls, lguess, gs = 4, [13, 2, 2, 2], 8, 4          # re-fed hits (level_offset 3), columns 12..14 of a W=15 window
    lay = O.StepLayout(ids=[0] * (n_input + sum(ls) + lguess), positions=[], n_input=n_input, level_sizes=ls, lguess=lguess, is_prefill=False, window=2)
    T = lay.T
    vis = O.dense_mask(lay, P, gs)
    S_max = 448
    q = torch.randn(T, H, d).bfloat16(); k = torch.randn(Hkv, S_max, d).bfloat16(); v = torch.randn(Hkv, S_max, d).bfloat16()
    ref = O.attention_dense(q.float().transpose(0, 1), k.float()[:, :P + T], v.float()[:, :P + T], vis).transpose(0, 1).reshape(T, H * d)
    for ns in (1, 4):
        out = ops.attn_fwd(q.reshape(T, -1).cuda(), k.cuda(), v.transpose(1, 2).contiguous().cuda(), ops.StepMask.from_levels(n_input, ls, lguess, gs, P),
                           H=H, Hkv=Hkv, d=d, n_splits=ns).float().cpu()
        assert torch.allclose(out, ref, atol=2e-2, rtol=2e-2), (ns, (out - ref).abs().max())


def test_attention_single_token_decode_and_empty_cache():
    import numpy as np
    import lade_oracle as O
    from lookaheaddecoding_amd import ops
    torch.manual_seed(2)
    H, d = 4, 64
    for (T, P) in ((1, 0), (1, 777), (3, 0), (64, 64), (65, 63)):
        q = torch.randn(T, H, d).half(); k = torch.randn(H, 896, d).half(); v = torch.randn(H, 896, d).half()
        vis = np.zeros((T, P + T), dtype=bool); vis[:, :P] = True; vis[:, P:] = np.tril(np.ones((T, T), dtype=bool))
        ref = O.attention_dense(q.float().transpose(0, 1), k.float()[:, :P + T], v.float()[:, :P + T], vis).transpose(0, 1).reshape(T, H * d)
        out = ops.attn_fwd(q.reshape(T, -1).cuda(), k.cuda(), v.transpose(1, 2).contiguous().cuda(), ops.StepMask(T=T, P=P, is_prefill=True), H=H, Hkv=H, d=d).float().cpu()
        assert torch.allclose(out, ref, atol=4e-3, rtol=1e-2), (T, P)


def test_pool_fullsize_lru_properties():
    """V = 32000, G = 15, gs = 4: capacity, LRU eviction order, move-to-end idempotence, untouched keys stay empty."""
    from lookaheaddecoding_amd.cabi import call, ptr
    V, G, gs = 32000, 15, 4
    pool_tok = torch.zeros(V, G, gs, dtype=torch.int32, device="cuda"); pool_cnt = torch.zeros(V, dtype=torch.int32, device="cuda")
    key = 31999
    grams = [[key, i, i + 1, i + 2, i + 3] for i in range(20)]           # 20 distinct n-grams under one key: the first 5 are evicted
    call("lade_pool_insert_ngrams", ptr(pool_tok), ptr(pool_cnt), V, G, gs, ptr(torch.tensor(grams, dtype=torch.int32, device="cuda")), len(grams))
    assert pool_cnt[key].item() == G and int(pool_cnt.sum()) == G
    assert pool_tok[key, :, 0].tolist() == list(range(5, 20))
    again = torch.tensor([grams[7]] * 3, dtype=torch.int32, device="cuda")  # re-inserting an entry moves it to the end, once
    call("lade_pool_insert_ngrams", ptr(pool_tok), ptr(pool_cnt), V, G, gs, ptr(again), 3)
    assert pool_tok[key, :, 0].tolist() == [5, 6] + list(range(8, 20)) + [7] and pool_cnt[key].item() == G
    go = torch.zeros(G * gs, dtype=torch.int32, device="cuda"); gn = torch.zeros(1, dtype=torch.int32, device="cuda")
    call("lade_pool_lookup", ptr(pool_tok), ptr(pool_cnt), V, G, gs, ptr(torch.tensor([key], dtype=torch.int32, device="cuda")), ptr(go), ptr(gn))
    assert gn.item() == G and go.view(G, gs)[-1].tolist() == [7, 8, 9, 10]
    call("lade_pool_lookup", ptr(pool_tok), ptr(pool_cnt), V, G, gs, ptr(torch.tensor([123], dtype=torch.int32, device="cuda")), ptr(go), ptr(gn))
    assert gn.item() == 0


def test_argmax_fullsize_and_odd_vocab():
    from lookaheaddecoding_amd import ops
    torch.manual_seed(3)
    for V in (32000, 32016, 32001):
        x = torch.randn(76, V, device="cuda").bfloat16()
        assert torch.equal(ops.argmax_rows(x).long().cpu(), torch.argmax(x.float(), dim=-1).cpu())


def test_skinny_gemm_vs_fp32_reference():
    from lookaheaddecoding_amd import ops
    torch.manual_seed(4)
    for (M, N, K) in ((60, 4096, 4096), (120, 12288, 4096), (76, 4096, 11008), (1, 512, 128), (37, 264, 192), (16, 2048, 2048), (31, 4096, 4096),
                      (240, 5120, 5120), (150, 4096, 4096), (180, 2048, 13824)):
        a = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        ref = a.float() @ w.float().t()
        for (S, bn, mb, mt, nt) in [x + (0,) for x in ((1, 128, 0, 0), (4, 128, 0, 0), (3, 64, 4, 1), (2, 256, 0, 0), (2, 192, 3, 1), (5, 64, 3, 1), (1, 256, 3, 1), (3, 64, 1, 1),
                                (2, 128, 1, 1), (4, 256, 1, 1), (2, 192, 2, 2), (3, 128, 2, 2), (1, 256, 2, 2), (2, 256, 4, 2), (3, 192, 4, 2), (1, 128, 4, 2),
                                (2, 256, 4, 4), (4, 192, 4, 4), (2, 192, 2, 1))] + [(2, 192, 2, 2, 2), (3, 256, 2, 2, 2), (1, 256, 2, 2, 4), (2, 192, 2, 2, 3),
                                    (2, 128, 1, 1, 2), (3, 256, 1, 1, 2), (2, 192, 3, 3, 2), (1, 256, 3, 3, 2), (2, 128, 3, 3, 0), (2, 192, 4, 4, 2),
                                    (3, 256, 4, 4, 2), (1, 128, 4, 4, 2), (2, 128, 4, 2, 2),
                                    # 192 / 256-row work-groups (steps of up to 256 tokens: the weights are streamed once)
                                    (1, 224, 2, 2, 1), (2, 224, 1, 1, 1), (3, 224, 4, 4, 1),      # 7-tile weight blocks
                                    (2, 128, 8, 4, 1), (4, 64, 8, 4, 1), (1, 128, 8, 4, 2), (3, 64, 8, 2, 1), (2, 128, 8, 2, 2), (2, 128, 6, 3, 1), (3, 64, 6, 3, 1),
                                    (2, 128, 6, 3, 2), (2, 64, 6, 2, 1), (1, 128, 6, 2, 2), (2, 128, 0, 0, 0)]:
            if K // 64 < S:
                continue
            out = ops.gemm_skinny(a, w, n_split=S, bn=bn, mb=mb, mt=mt, nt=nt).float()
            assert torch.allclose(out, ref, atol=2e-2 * max(1.0, ref.abs().max().item() / 8), rtol=2e-2), (M, N, K, S, bn, mb, mt, nt, (out - ref).abs().max())


@pytest.mark.parametrize("shape,layers,W,N,G", [("llama2-7b", 2, 15, 5, 15), ("codellama-13b", 2, 20, 7, 20), ("llama2-70b", 1, 15, 5, 15)])
def test_fullsize_shapes_end_to_end_on_a_successor_model(shape, layers, W, N, G):
    """BASELINE shapes (hidden / heads / GQA / vocab at full size, few layers) through the whole step in bf16, graph and
    eager mode: a model whose greedy continuation is known in closed form (o_proj / down_proj zeroed, lm_head = shifted
    embedding -> token t is followed by (t+1) mod C) must be decoded exactly, with every step accepting a full n-gram,
    and lookahead must equal plain greedy decoding on the same engine."""
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    from lookaheaddecoding_amd.weights import make_config, random_weights_torch
    cfg = make_config(shape)
    cfg["layers"] = layers
    w = random_weights_torch(cfg, seed=0, dtype=torch.bfloat16, device="cuda")
    eng = StepEngine(cfg, w, dtype=torch.bfloat16, device="cuda", max_seq=1024, max_T=512)
    del w
    C = 256
    eng.zero_projections(("wo", "wd"))
    head = eng.embed.clone()
    head[:C] = eng.embed[(torch.arange(C, device="cuda") - 1) % C]
    eng.lm_head = head
    prompt = [i % C for i in range(300)]
    n_new = 96
    want = [(prompt[-1] + 1 + i) % C for i in range(n_new)]
    assert eng.plain_greedy(prompt, len(prompt) + 24)[len(prompt):] == want[:24]
    for use_graph in (False, True):
        dec = LookaheadDecoder(eng, W, N, G, pool_from_prompt=True, use_graph=use_graph)
        out = dec.greedy(prompt, len(prompt) + n_new, rng=random.Random(1))
        assert out.tokens[len(prompt):] == want, (shape, use_graph)
        assert out.steps <= n_new // (N - 1) + N, (shape, use_graph, out.steps)      # ~N-1 tokens per step after the window has filled
