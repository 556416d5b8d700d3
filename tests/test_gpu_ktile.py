"""K-tile-major weights for the decode GEMMs (`lade_weight_to_ktile`, `lade_gemm_skinny_kt`, the engine's second weight copy).

The layout changes only the addresses the weight DMA reads: the arithmetic and its order are those of `lade_gemm_skinny`, so every
comparison here is BIT-EXACT against the row-major path, which the other GPU tests compare with torch / the fp32 oracle.
Reference: the nn.Linear projections of the step, lade/models/modeling_llama.py:360-380, 492-494, 558."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from lookaheaddecoding_amd import cabi
from lookaheaddecoding_amd.weights import make_config, random_weights_numpy


def test_weight_to_ktile_is_the_permutation_and_refuses_bad_shapes():
    from lookaheaddecoding_amd import cabi, ops
    torch.manual_seed(0)
    for dtype in (torch.bfloat16, torch.float16):
        for (N, K) in ((160, 320), (8, 64), (1000, 192), (4096, 11008)):
            w = torch.randn(N, K, device="cuda").to(dtype)
            assert torch.equal(ops.to_ktile(w), w.view(N, K // 64, 64).permute(1, 0, 2).contiguous()), (N, K, dtype)
            assert torch.equal(ops.from_ktile(ops.to_ktile(w)), w)                    # and back
            scratch = torch.full((N, K + 64), 7.0, dtype=dtype, device="cuda")        # into a wider scratch: the padding is not touched
            ops.from_ktile(ops.to_ktile(w), out=scratch[:, :K])
            assert torch.equal(scratch[:, :K], w) and bool((scratch[:, K:] == 7.0).all())
        w = torch.randn(96, 256, device="cuda").to(dtype)[:, :192]                  # a row stride larger than K
        assert torch.equal(ops.to_ktile(w), w.reshape(96, 3, 64).permute(1, 0, 2).contiguous())
    w = torch.randn(32, 96, device="cuda").bfloat16()
    out = torch.empty(2, 32, 64, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(cabi.LadeHipError):                                          # K % 64 != 0
        cabi.call("lade_weight_to_ktile", cabi.ptr(w), 96, cabi.ptr(out), 32, 96, cabi.dtype_code(w))
    w = torch.randn(32, 128, device="cuda").bfloat16()
    with pytest.raises(cabi.LadeHipError):                                          # in place
        cabi.call("lade_weight_to_ktile", cabi.ptr(w), 128, cabi.ptr(w), 32, 128, cabi.dtype_code(w))
    with pytest.raises(cabi.LadeHipError):                                          # fp32 has no skinny GEMM
        cabi.call("lade_weight_to_ktile", cabi.ptr(w), 128, cabi.ptr(out), 32, 128, 2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_on_ktile_weights_is_bit_identical_to_row_major(dtype):
    """every row class (32 ... 256 rows), split-K partials and the direct output, weight rows that do not fill the last work-group
    (the clamped tail), one K tile per split, 7B-sized strides"""
    from lookaheaddecoding_amd import ops
    torch.manual_seed(1)
    cases = [  # M, N, K, (bn, mb, mt, nt), splits
        (30, 1000, 512, (128, 1, 1, 0), (1, 2, 4)),
        (60, 4096, 4096, (128, 2, 2, 0), (1, 8)),
        (60, 1000, 1024, (96, 2, 1, 1), (1, 2)),
        (92, 264, 256, (64, 3, 1, 0), (1, 2, 4)),
        (128, 776, 512, (192, 4, 2, 0), (1, 3)),
        (180, 520, 384, (128, 6, 3, 1), (1, 2)),
        (240, 1032, 640, (64, 8, 4, 1), (1, 5)),
        (1, 12288, 4096, (96, 1, 1, 1), (2,)),
        (7, 64, 64, (64, 1, 1, 0), (1,)),
    ]
    for (M, N, K, (bn, mb, mt, nt), splits) in cases:
        a = torch.randn(M, K, device="cuda").to(dtype)
        w = (torch.randn(N, K, device="cuda") * 0.05).to(dtype)
        wk = ops.to_ktile(w)
        for S in splits:
            if S > 1:
                p0 = torch.full((S * M * N,), float("nan"), dtype=torch.float32, device="cuda")
                p1 = torch.full((S * M * N,), float("nan"), dtype=torch.float32, device="cuda")
                ops.gemm_parts(a, w, p0, S, bn, mb, mt, nt)
                ops.gemm_parts(a, wk, p1, S, bn, mb, mt, nt)
                assert torch.isfinite(p0).all() and torch.equal(p0, p1), (M, N, K, S)
            o0 = ops.gemm_skinny(a, w, n_split=S, bn=bn, mb=mb, mt=mt, nt=nt)
            o1 = ops.gemm_skinny(a, wk, n_split=S, bn=bn, mb=mb, mt=mt, nt=nt)
            assert torch.equal(o0, o1), (M, N, K, S)
            ref = a.float() @ w.float().t()
            assert torch.allclose(o1.float(), ref, atol=0.02 * K ** 0.5 * 0.05 + 0.02, rtol=2e-2), (M, N, K, S, (o1.float() - ref).abs().max())


def test_swiglu_epilogue_on_ktile_weights_is_bit_identical():
    from lookaheaddecoding_amd import ops
    torch.manual_seed(2)
    for (M, inter, K, bn, mt) in ((60, 11008, 4096, 96, 1), (30, 1408, 512, 128, 1), (120, 1792, 1024, 96, 2), (16, 256, 128, 64, 1)):
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = ops.interleave_gate_up((torch.randn(inter, K, device="cuda") * 0.05).bfloat16(), (torch.randn(inter, K, device="cuda") * 0.05).bfloat16())
        mb = 1 if M <= 32 else 2 if M <= 64 else 3 if M <= 96 else 4
        o0 = torch.full((M, inter), float("nan"), dtype=torch.bfloat16, device="cuda")
        o1 = torch.full((M, inter), float("nan"), dtype=torch.bfloat16, device="cuda")
        ops.gemm_swiglu(a, w, o0, bn, mb, mt, 1)
        ops.gemm_swiglu(a, ops.to_ktile(w), o1, bn, mb, mt, 1)
        assert torch.isfinite(o0.float()).all() and torch.equal(o0, o1), (M, inter, K)


def _engine(monkeypatch, ktile, cfg, w, **kw):
    from lookaheaddecoding_amd.engine import StepEngine
    monkeypatch.setenv("LADE_W_KTILE", ktile)
    return StepEngine(cfg, w, dtype=torch.bfloat16, max_seq=512, max_T=256, **kw)


def test_engine_with_and_without_the_ktile_copy_gives_the_same_logits_bit_for_bit(monkeypatch):
    """two engines on the same weights, one streaming the row-major weights and one the K-tile-major copies, on the SAME GEMM
    configurations (rank 0's table adopted, as lookahead-parallel ranks do): logits and the appended K/V rows are bit-identical, for a
    lookahead-shaped step, a one-token step and a 240-row step; zero_projections reaches both layouts"""
    from lookaheaddecoding_amd import ops
    cfg = make_config("tiny-d128", max_pos=512)
    w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=4, std=0.05).items()}
    e0 = _engine(monkeypatch, "0", cfg, w)
    e1 = _engine(monkeypatch, "1", cfg, w)
    assert not e0.ktile and e1.ktile and all(n + "_kt" in lw for lw in e1.layers for n in e1.LAYER_GEMMS)
    assert not any(k.endswith("_kt") for lw in e0.layers for k in lw)
    e1.adopt_gemm_cfg(e0.tune_all())
    g = torch.Generator().manual_seed(5)
    P = 40
    prompt = torch.randint(3, cfg["vocab"], (P,), generator=g).tolist()
    for T in (60, 1, 240):
        ids = torch.randint(3, cfg["vocab"], (T,), generator=g).to(torch.int32).cuda()
        pos = (P + torch.arange(T)).to(torch.int32).cuda()
        sel = torch.arange(T, dtype=torch.int32).cuda()
        outs = []
        for e in (e0, e1):
            e.reset()
            e.prefill(prompt, rows=[P - 1])
            lg = e.forward(ids, pos, ops.StepMask(T=T, P=P, is_prefill=True), sel, T)
            outs.append((lg.clone(), e.k_cache(0)[:, P:P + T].clone(), e.vt_cache(cfg["layers"] - 1)[:, :, P:P + T].clone()))
        for x, y in zip(*outs):
            assert torch.isfinite(x.float()).all() and torch.equal(x, y), T
    # a replaced output projection reaches the copy the step streams
    before = outs[1][0]
    head = (torch.randn(cfg["vocab"], cfg["hidden"], generator=g) * 0.05).to(torch.bfloat16).cuda()
    res = []
    for e in (e0, e1):
        e.lm_head = head
        e.reset()
        e.prefill(prompt, rows=[P - 1])
        res.append(e.forward(ids, pos, ops.StepMask(T=T, P=P, is_prefill=True), sel, T).clone())
    assert torch.equal(res[0], res[1]) and not torch.equal(res[1], before)
    assert e1._lm_kt is not None and torch.equal(e1._lm_kt, ops.to_ktile(head)) and e0._lm_kt is None
    e1.zero_projections(("wo", "wd"))
    assert all(float(lw[n].abs().sum()) == 0.0 for lw in e1.layers for n in ("wo", "wd", "wo_kt", "wd_kt"))
    assert all(float(lw["wqkv_kt"].abs().sum()) > 0.0 for lw in e1.layers)


def test_engine_holding_the_weights_ktile_only_gives_the_same_logits_and_prefills_through_the_library(monkeypatch):
    """LADE_W_KTILE=only (what a model too large to be held twice gets): no row-major projection weights are kept; decode-width steps are
    bit-identical to the row-major engine on the same GEMM configurations, and a prefill chunk wider than 256 rows - library GEMM on the
    row-major operand rebuilt by lade_weight_from_ktile - is bit-identical too"""
    from lookaheaddecoding_amd import ops
    cfg = make_config("tiny-d128", max_pos=1024)
    w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=6, std=0.05).items()}
    from lookaheaddecoding_amd.engine import StepEngine
    monkeypatch.setenv("LADE_W_KTILE", "0")
    e0 = StepEngine(cfg, w, dtype=torch.bfloat16, max_seq=1024, max_T=512)
    monkeypatch.setenv("LADE_W_KTILE", "only")
    e2 = StepEngine(cfg, w, dtype=torch.bfloat16, max_seq=1024, max_T=512)
    assert e2.ktile and e2.ktile_only and not any(n in lw for lw in e2.layers for n in e2.LAYER_GEMMS)
    assert all(n + "_kt" in lw for lw in e2.layers for n in e2.LAYER_GEMMS)
    e2.adopt_gemm_cfg(e0.tune_all())
    g = torch.Generator().manual_seed(7)
    prompt = torch.randint(3, cfg["vocab"], (300,), generator=g).tolist()           # one causal chunk of 300 rows: wider than the skinny GEMM takes
    P, T = len(prompt), 60
    ids = torch.randint(3, cfg["vocab"], (T,), generator=g).to(torch.int32).cuda()
    pos = (P + torch.arange(T)).to(torch.int32).cuda()
    sel = torch.arange(T, dtype=torch.int32).cuda()
    outs = []
    for e in (e0, e2):
        lg_p, done = e.prefill(prompt, rows=[P - 2, P - 1])
        assert done == 0
        lg = e.forward(ids, pos, ops.StepMask(T=T, P=P, is_prefill=True), sel, T)
        outs.append((lg_p.clone(), lg.clone(), e.k_cache(1)[:, :P + T].clone()))
    for x, y in zip(*outs):
        assert torch.isfinite(x.float()).all() and torch.equal(x, y)
    # its own autotune never picks the library for a decode-width step (there is no row-major operand to give it)
    monkeypatch.setenv("LADE_W_KTILE", "only")
    e3 = StepEngine(cfg, w, dtype=torch.bfloat16, max_seq=1024, max_T=512)
    e3.prefill(prompt[:40], rows=[39])
    lg3 = e3.forward(ids, (40 + torch.arange(T)).to(torch.int32).cuda(), ops.StepMask(T=T, P=40, is_prefill=True), sel, T)
    assert torch.isfinite(lg3.float()).all()
    e2.zero_projections(("wo", "wd"))
    assert all(float(lw[n + "_kt"].abs().sum()) == 0.0 for lw in e2.layers for n in ("wo", "wd"))


def test_ktile_layout_decision_dual_when_it_fits_only_when_it_does_not(monkeypatch):
    from lookaheaddecoding_amd.engine import StepEngine
    cfg = make_config("tiny-d128", max_pos=512)
    w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=4, std=0.05).items()}
    e = _engine(monkeypatch, "auto", cfg, w)
    assert e.ktile and e.ktile_bytes == sum(lw[n].numel() * 2 for lw in e.layers for n in e.LAYER_GEMMS) + e.lm_head.numel() * 2
    monkeypatch.setattr(StepEngine, "KTILE_RESERVE", 1 << 50)                      # "a second copy does not fit" (Llama-2-70B): K-tile-major only
    e = _engine(monkeypatch, "auto", cfg, w)
    assert e.ktile and e.ktile_only and e.ktile_bytes == e.lm_head.numel() * 2
    e = _engine(monkeypatch, "0", cfg, w)
    assert not e.ktile and e.ktile_bytes == 0 and all(n in lw for lw in e.layers for n in e.LAYER_GEMMS)
    e = StepEngine(cfg, w, dtype=torch.float32, max_seq=256, max_T=64)              # fp32 has no skinny GEMM: nothing to copy
    assert not e.ktile


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_ring_depth_is_a_launch_parameter_and_never_changes_a_bit(dtype):
    """the LDS ring depth (C ABI `ring`, chosen per projection by the engine's autotune) changes how many tiles are in flight, not the
    arithmetic: every compiled depth (2, 3, 4, 5, 6, 8) that fits the LDS gives the bits of the default, for split-K partials, the direct
    output, the SwiGLU epilogue and a K range shorter than the ring; a ring that does not fit (or is not compiled: 7, 9) is refused"""
    from lookaheaddecoding_amd import ops
    torch.manual_seed(3)
    for (M, N, K, (bn, mb, mt, nt), S) in [(60, 1000, 1024, (96, 2, 1, 1), 2), (60, 512, 4096, (128, 2, 2, 0), 8), (120, 776, 512, (192, 4, 2, 0), 1),
                                           (30, 264, 128, (64, 1, 1, 0), 1), (240, 520, 640, (128, 8, 4, 1), 2)]:
        a = torch.randn(M, K, device="cuda").to(dtype)
        wk = ops.to_ktile((torch.randn(N, K, device="cuda") * 0.05).to(dtype))
        ref = ops.gemm_skinny(a, wk, n_split=S, bn=bn, mb=mb, mt=mt, nt=nt)
        fits = [r for r in (2, 3, 4, 5, 6, 8) if r * (bn + 32 * mb) * 128 <= 160 * 1024]
        assert fits
        for ring in fits:
            assert torch.equal(ops.gemm_skinny(a, wk, n_split=S, bn=bn, mb=mb, mt=mt, nt=nt, ring=ring), ref), (M, N, K, S, ring)
        for ring in set(range(2, 9)) - set(fits):
            with pytest.raises(cabi.LadeHipError):
                ops.gemm_skinny(a, wk, n_split=S, bn=bn, mb=mb, mt=mt, nt=nt, ring=ring)
    with pytest.raises(cabi.LadeHipError):
        ops.gemm_skinny(a, wk, n_split=1, bn=128, mb=8, mt=4, nt=1, ring=9)
    # SwiGLU epilogue
    inter, hid, M = 352, 256, 60
    wg, wu = [(torch.randn(inter, hid, device="cuda") * 0.05).to(dtype) for _ in range(2)]
    wf = ops.to_ktile(ops.interleave_gate_up(wg, wu))
    a = torch.randn(M, hid, device="cuda").to(dtype)
    o = [torch.empty(M, inter, dtype=dtype, device="cuda") for _ in range(3)]
    ops.gemm_swiglu(a, wf, o[0], 96, 2, 1, 1)
    ops.gemm_swiglu(a, wf, o[1], 96, 2, 1, 1, ring=3)
    ops.gemm_swiglu(a, wf, o[2], 96, 2, 1, 1, ring=8)
    assert torch.equal(o[0], o[1]) and torch.equal(o[0], o[2])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_wider_than_256_rows_runs_as_row_blocks_on_every_tile_shape(dtype):
    """M > 256 (the reference's default W = 60, N = 8, G = 60 feeds 847 rows; prefill chunks 2304): several 256-row blocks per launch.
    Every output element is one K-ordered MFMA accumulation whatever the tile shape, so the 128-row-wide tiles, the compute-shaped
    256 x 256 tile (double-buffered) and the <= 256-row launches of the decode step all give the same bits; split-K partials of the
    256 x 256 tile (stored straight from the accumulators) equal those of the staged shapes; and the result is the fp32 product"""
    from lookaheaddecoding_amd import ops
    torch.manual_seed(4)
    for (M, N, K) in [(300, 520, 256), (512, 1024, 1024), (847, 776, 512), (2304, 264, 384)]:
        a = torch.randn(M, K, device="cuda").to(dtype)
        w = (torch.randn(N, K, device="cuda") * 0.05).to(dtype)
        wk = ops.to_ktile(w)
        base = torch.cat([ops.gemm_skinny(a[r0:r0 + 240], wk, n_split=1, bn=128, mb=8, mt=4, nt=1) for r0 in range(0, M, 240)])
        for (bn, mt, nt) in [(128, 4, 1), (128, 2, 2), (64, 4, 1), (256, 4, 2), (256, 2, 4)]:
            for W_ in (w, wk):
                got = ops.gemm_skinny(a, W_, n_split=1, bn=bn, mb=8, mt=mt, nt=nt)
                assert torch.equal(got, base), (M, N, K, bn, mt, nt, W_.dim())
        ref = a.float() @ w.float().t()
        assert torch.allclose(base.float(), ref, atol=0.02 * K ** 0.5 * 0.05 + 0.02, rtol=2e-2)
        p0 = torch.full((2 * M * N,), float("nan"), dtype=torch.float32, device="cuda")
        p1 = torch.full((2 * M * N,), float("nan"), dtype=torch.float32, device="cuda")
        ops.gemm_parts(a, wk, p0, 2, 128, 8, 4, 1)
        ops.gemm_parts(a, wk, p1, 2, 256, 8, 4, 2)
        assert torch.isfinite(p0).all() and torch.equal(p0, p1), (M, N, K)


def test_weight_memory_plan_keeps_row_major_originals_while_the_budget_lasts_and_skips_aliased_projections(monkeypatch):
    """what `_build_ktile_copies` decides when a second copy of everything does not fit (Llama-2-70B): owned projections are converted
    and their row-major originals kept for the first layers only (the budget), released for the rest - prefill then rebuilds the
    operand for those layers alone; projections whose row-major weight is the CALLER's storage (HF drop-in: o / down) are neither
    released nor copied.  The logits of a decode-width step and of a library prefill chunk are bit-identical to the all-row-major engine;
    `refresh_ktile` carries an in-place edit of an aliased weight into the copy the decode steps stream."""
    from lookaheaddecoding_amd import ops
    from lookaheaddecoding_amd.engine import StepEngine
    cfg = make_config("tiny-d128", max_pos=1024)
    w_cpu = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=8, std=0.05).items()}
    w_gpu = {k: v.to("cuda", torch.bfloat16) for k, v in w_cpu.items()}            # an HF-like caller: the tensors already live on the GPU
    nbytes = lambda t: t.numel() * t.element_size()
    monkeypatch.setenv("LADE_W_KTILE", "0")
    e0 = StepEngine(cfg, w_gpu, dtype=torch.bfloat16, max_seq=1024, max_T=512)
    assert {n for (_l, n) in e0._aliased} == {"wo", "wd"} and len(e0._aliased) == 2 * cfg["layers"]
    per_layer = sum(nbytes(e0.layers[0][n]) for n in ("wqkv", "wgu"))
    scratch = per_layer
    # budget: the scratch + ONE layer's originals (+ a margin far below a second layer's)
    monkeypatch.setenv("LADE_W_KTILE", "auto")
    monkeypatch.setenv("LADE_KTILE_BUDGET_MB", repr((scratch + per_layer + 4096) / (1 << 20)))
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    e1 = StepEngine(cfg, w_gpu, dtype=torch.bfloat16, max_seq=1024, max_T=512)
    assert e1.kt_names == ("wqkv", "wgu") and e1.ktile and not e1.ktile_only
    assert all("wo_kt" not in lw and "wd_kt" not in lw and lw["wo"].data_ptr() == w_gpu[f"layers.{i}.wo"].data_ptr() for i, lw in enumerate(e1.layers))
    assert [("wqkv" in lw, "wgu" in lw) for lw in e1.layers] == [(True, True)] + [(False, False)] * (cfg["layers"] - 1)
    assert e1.rows_kept == 2 and e1.rows_total == 2 * cfg["layers"]
    # an owned-everything engine under the same budget converts all four projections
    e2 = StepEngine(cfg, w_cpu, dtype=torch.bfloat16, max_seq=1024, max_T=512)
    assert e2.kt_names == e2.LAYER_GEMMS and not e2._aliased
    monkeypatch.delenv("LADE_KTILE_BUDGET_MB")
    table = e0.tune_all()
    for e in (e1, e2):
        e.adopt_gemm_cfg(table)
    g = torch.Generator().manual_seed(9)
    prompt = torch.randint(3, cfg["vocab"], (300,), generator=g).tolist()
    P, T = len(prompt), 60
    ids = torch.randint(3, cfg["vocab"], (T,), generator=g).to(torch.int32).cuda()
    pos = (P + torch.arange(T)).to(torch.int32).cuda()
    sel = torch.arange(T, dtype=torch.int32).cuda()
    outs = []
    for e in (e0, e1, e2):
        lg_p, _ = e.prefill(prompt, rows=[P - 1])
        outs.append((lg_p.clone(), e.forward(ids, pos, ops.StepMask(T=T, P=P, is_prefill=True), sel, T).clone()))
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1])
    # in-place update of a weight the dual engine aliases / copies: refresh_ktile() re-derives the streamed copies
    monkeypatch.setenv("LADE_W_KTILE", "1")
    e3 = StepEngine(cfg, w_gpu, dtype=torch.bfloat16, max_seq=1024, max_T=512)
    assert e3.kt_names == e3.LAYER_GEMMS and e3.rows_kept == e3.rows_total
    w_gpu["layers.0.wo"].mul_(0.5)
    assert not torch.equal(e3.layers[0]["wo_kt"], ops.to_ktile(e3.layers[0]["wo"]))
    e3.refresh_ktile()
    assert all(torch.equal(lw[n + "_kt"], ops.to_ktile(lw[n])) for lw in e3.layers for n in e3.LAYER_GEMMS)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_argmax_epilogue_returns_the_ids_of_argmax_over_the_materialised_logits(dtype):
    """lade_gemm_skinny(epilogue = 2) + lade_argmax_pairs against lade_argmax_rows on the logits the same kernel writes with epilogue 0
    (torch.argmax of lade/decoding.py:1021 on the lm_head rows of modeling_llama.py:1541): identical ids - including TIES, which the
    16-bit rounding of the logits makes common at vocabulary width (duplicated weight rows force them here; the lowest column wins) -
    for every column-block width and wave grid the lm_head tuner may pick, row-major and K-tile-major weights, vocabulary sizes that
    are not a multiple of the block, 1 / 16 / 31 / 60 / 100 rows."""
    from lookaheaddecoding_amd import ops
    torch.manual_seed(3)
    K = 256
    for V in (32000, 32016, 1000):
        w = (torch.randn(V, K, device="cuda") * 0.05).to(dtype)
        w[V // 2] = w[7]                                   # exact ties between columns of different blocks ...
        w[8] = w[7]                                        # ... and inside one MFMA tile
        w[V - 1] = w[V - 3]                                # ... and in the ragged last block
        wkt = ops.to_ktile(w)
        for M in (1, 16, 31, 60, 100):
            a = torch.randn(M, K, device="cuda").to(dtype)
            a[0] = w[7].float() * 40                       # row 0's best columns are the tied ones (7, 8, V/2): 7 must win
            mb = (M + 31) // 32
            for bn in (64, 96, 128, 192, 256):
                for nt in (0, 1, 2):
                    for ww in (w, wkt):
                        try:
                            logits = ops.gemm_skinny(a, ww, n_split=1, bn=bn, mb=mb, nt=nt)
                        except cabi.LadeHipError:
                            continue                       # wave grid not in the shape table
                        want = ops.argmax_rows(logits)
                        got = torch.full((M,), -1, dtype=torch.int32, device="cuda")
                        ops.gemm_argmax(a, ww, got, bn=bn, mb=mb, nt=nt)
                        assert torch.equal(got, want), (V, M, bn, nt, ww.dim(), (got != want).nonzero().flatten().tolist()[:5])
            assert int(want[0]) == 7
    with pytest.raises(cabi.LadeHipError):                 # the argmax epilogue has no split-K form
        cabi.call("lade_gemm_skinny", cabi.ptr(a), K, cabi.ptr(w), K, None, 0, cabi.ptr(torch.empty(64, device="cuda")), 1, 1000, K, 2, 128, 1, 0, 0, 0, 2,
                  cabi.dtype_code(a))


def test_argmax_nan_rule_is_torchs_in_all_three_kernels():
    """torch.argmax treats NaN as the maximum (the first NaN of a row wins).  lade_argmax_rows, the lm_head GEMM's argmax epilogue and
    lade_argmax_pairs follow the same rule, so that a step with the fused tail surfaces a NaN logit exactly where the unfused one does
    (round-4 advice: the epilogue's `v > best` never selected a NaN).  And the C ABI refuses an argmax epilogue whose column block it would
    have to narrow silently (the pair buffer's stride is the caller's ceil(N / bn))."""
    from lookaheaddecoding_amd import ops
    torch.manual_seed(5)
    K, V, M = 256, 1000, 40
    w = (torch.randn(V, K, device="cuda") * 0.05).bfloat16()
    a = torch.randn(M, K, device="cuda").bfloat16()
    a[3, 5] = float("nan")                                 # row 3: every logit is NaN -> column 0
    w[700, 9] = float("nan")                               # column 700 is NaN for every (finite) row
    w[123, 9] = float("nan")                               # ... and so is column 123: the lower one wins
    for ww in (w, ops.to_ktile(w)):
        logits = ops.gemm_skinny(a, ww, n_split=1, bn=128, mb=2)
        want_t = torch.argmax(logits.float(), dim=-1).to(torch.int32)
        assert int(want_t[3]) == 0 and int(want_t[0]) == 123
        assert torch.equal(ops.argmax_rows(logits), want_t)
        for bn in (64, 96, 128, 256):
            got = torch.full((M,), -1, dtype=torch.int32, device="cuda")
            ops.gemm_argmax(a, ww, got, bn=bn, mb=2)
            assert torch.equal(got, want_t), (bn, ww.dim())
    lf = torch.randn(5, 333, device="cuda")
    lf[2, 40] = float("nan"); lf[2, 17] = float("nan"); lf[4, 0] = float("inf")
    assert torch.equal(ops.argmax_rows(lf), torch.argmax(lf, dim=-1).to(torch.int32))
    with pytest.raises(cabi.LadeHipError, match="epilogue 2"):
        cabi.call("lade_gemm_skinny", cabi.ptr(a), K, cabi.ptr(w), K, None, 0, cabi.ptr(torch.empty(4096, device="cuda")), M, V, K, 1, 256, 6, 0, 0, 0, 2,
                  cabi.dtype_code(a))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_embed_rmsnorm_is_the_row_gather_followed_by_the_norm_bit_for_bit(dtype):
    """lade_embed_rmsnorm (embedding lookup inside the first layer's input norm; modeling_llama.py:1413 + :857) against
    lade_gather_rows + lade_rmsnorm: the same residual rows and the same normed rows, ids outside the table clamped like the gather."""
    from lookaheaddecoding_amd import ops
    torch.manual_seed(4)
    for hidden in (64, 4096, 5120, 8192):
        table = torch.randn(1000, hidden, device="cuda").to(dtype)
        wn = (1 + 0.1 * torch.randn(hidden, device="cuda")).to(dtype)
        ids = torch.randint(0, 1000, (70,), device="cuda", dtype=torch.int32)
        ids[3], ids[4] = -5, 4000
        for rows in (1, 60, 70):
            x0 = torch.empty(rows, hidden, dtype=dtype, device="cuda")
            ops.gather_rows(table, ids, out=x0, rows=rows)
            h0 = ops.rmsnorm(x0, wn, 1e-5)
            x1 = torch.full((rows + 1, hidden), 3.0, dtype=dtype, device="cuda")
            h1 = torch.full((rows + 1, hidden), 3.0, dtype=dtype, device="cuda")
            ops.embed_rmsnorm(table, ids, x1[:rows], wn, 1e-5, h1[:rows], rows)
            assert torch.equal(x1[:rows], x0) and torch.equal(h1[:rows], h0), (hidden, rows)
            assert bool((x1[rows] == 3.0).all()) and bool((h1[rows] == 3.0).all())          # nothing written past the rows


def test_step_with_the_fused_tail_gives_the_argmax_and_cache_rows_of_the_unfused_step(monkeypatch):
    """LADE_DEBUG=fuse_tail=0 | 1 (default on): embedding lookup inside the first norm + argmax inside the lm_head GEMM.  One engine, the same
    step with the switch off and on: the same argmax ids as argmax_rows over the logits of the unfused step, the same appended K/V rows,
    for a lookahead-shaped step (31 logits rows of 60), a one-token step and a step whose logits rows span 100 rows."""
    from lookaheaddecoding_amd import ops
    cfg = make_config("tiny-d128", max_pos=512)
    w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=4, std=0.05).items()}
    e = _engine(monkeypatch, "1", cfg, w)
    g = torch.Generator().manual_seed(6)
    P = 40
    prompt = torch.randint(3, cfg["vocab"], (P,), generator=g).tolist()
    for T, n_sel in ((60, 31), (1, 1), (120, 100)):
        ids = torch.randint(3, cfg["vocab"], (T,), generator=g).to(torch.int32).cuda()
        pos = (P + torch.arange(T)).to(torch.int32).cuda()
        sel = torch.arange(T - n_sel, T, dtype=torch.int32).cuda()
        outs = []
        for fuse in ("0", "1"):
            monkeypatch.setenv("LADE_DEBUG", f"fuse_tail={fuse}")
            e.reset()
            e.prefill(prompt, rows=[P - 1])
            am = torch.full((n_sel,), -1, dtype=torch.int32, device="cuda")
            lg = e.forward(ids, pos, ops.StepMask(T=T, P=P, is_prefill=True), sel, n_sel, argmax_out=am)
            assert lg is None
            outs.append((am.clone(), e.k_cache(0)[:, P:P + T].clone(), e.vt_cache(cfg["layers"] - 1)[:, :, P:P + T].clone()))
        monkeypatch.setenv("LADE_DEBUG", "fuse_tail=0")
        e.reset()
        e.prefill(prompt, rows=[P - 1])
        want = ops.argmax_rows(e.forward(ids, pos, ops.StepMask(T=T, P=P, is_prefill=True), sel, n_sel))
        assert torch.equal(outs[0][0], want) and all(torch.equal(x, y) for x, y in zip(*outs)), (T, n_sel)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_160_row_class_is_bit_identical_to_the_192_row_class_on_every_shape(dtype):
    """Round 6: a 129..160-row step runs five m-blocks per wave ((1,5,n,nt) wave grids, up to 256 weight rows per work-group) instead of
    padding to 192 rows.  Same K slices, same MFMA chain per output element: split-K partials, the direct output (staged and - for the
    256-row-wide tile whose fp32 staging exceeds the LDS - stored straight from the accumulators), the SwiGLU epilogue and every ring depth
    must equal the 192-row class BIT FOR BIT; weight rows that do not fill the last work-group, row-major and K-tile-major."""
    from lookaheaddecoding_amd import ops
    torch.manual_seed(11)
    shapes = ((64, 1), (96, 1), (128, 1), (192, 1), (256, 1), (128, 2), (256, 2))          # (bn, nt) -> (1,5,bn/32/nt,nt)
    for (M, N, K) in ((129, 1000, 512), (150, 4096, 1024), (160, 776, 384), (156, 264, 256)):
        a = torch.randn(M, K, device="cuda").to(dtype)
        w = (torch.randn(N, K, device="cuda") * 0.05).to(dtype)
        wk = ops.to_ktile(w)
        ref = a.float() @ w.float().t()
        for S in (1, 2, 4):
            if K // 64 < 2 * S:
                continue
            want = ops.gemm_skinny(a, w, n_split=S, bn=128, mb=6, mt=3, nt=1)
            p_want = None
            if S > 1:
                p_want = torch.empty(S * M * N, dtype=torch.float32, device="cuda")
                ops.gemm_parts(a, w, p_want, S, 128, 6, 3, 1)
            assert torch.allclose(want.float(), ref, atol=0.02 * K ** 0.5 * 0.05 + 0.02, rtol=2e-2)
            for (bn, nt) in shapes:
                for ring in (0, 2, 3):
                    if ring * (bn + 160) * 128 > 160 * 1024:
                        continue
                    for wt in (w, wk):
                        got = ops.gemm_skinny(a, wt, n_split=S, bn=bn, mb=5, mt=5, nt=nt, ring=ring)
                        assert torch.equal(got, want), (M, N, K, S, bn, nt, ring, wt.dim())
                    if S > 1:
                        p = torch.full((S * M * N,), float("nan"), dtype=torch.float32, device="cuda")
                        ops.gemm_parts(a, wk, p, S, bn, 5, 5, nt, ring)
                        assert torch.equal(p, p_want), (M, N, K, S, bn, nt, ring)
    # SwiGLU epilogue (unsplit gate/up) and mb = 0 (the row class by M)
    for (M, inter, K, bn) in ((150, 1792, 1024, 96), (132, 1408, 512, 128), (160, 256, 128, 64)):
        a = torch.randn(M, K, device="cuda").to(dtype)
        w = ops.interleave_gate_up((torch.randn(inter, K, device="cuda") * 0.05).to(dtype), (torch.randn(inter, K, device="cuda") * 0.05).to(dtype))
        want = ops.gemm_swiglu(a, w, torch.empty(M, inter, dtype=dtype, device="cuda"), 128 if bn != 64 else 64, 6, 3 if bn != 64 else 2, 1)
        got = ops.gemm_swiglu(a, ops.to_ktile(w), torch.empty(M, inter, dtype=dtype, device="cuda"), bn, 5, 5, 1)
        assert torch.equal(got, want), (M, inter, K, bn)
        assert torch.equal(ops.gemm_skinny(a, w, bn=128), ops.gemm_skinny(a, w, bn=128, mb=5, mt=5, nt=1)), "mb = 0 picks the 160-row class for 129..160 rows"


def test_experimental_gemm_structures_are_refused_by_the_default_build_and_correct_in_the_experimental_one():
    """register-resident activations (lade_gemm_ra_kt) and the ping-pong K loop (ring >= 10): built by `make EXPERIMENTAL=1` only (both measured
    slower or equal: DESIGN 4.9).  Default build: an error code, never a silent fallback.  Experimental build: RA bit-identical to the ring
    kernel's partials, ping-pong equal to the product within fp32 summation order."""
    from lookaheaddecoding_amd import ops
    torch.manual_seed(12)
    M, N, K, S = 120, 1024, 1024, 2
    a = torch.randn(M, K, device="cuda").bfloat16()
    wk = ops.to_ktile((torch.randn(N, K, device="cuda") * 0.05).bfloat16())
    part = torch.empty(S * M * N, dtype=torch.float32, device="cuda")
    if not cabi.experimental():
        with pytest.raises(cabi.LadeHipError, match="not in this build"):
            ops.gemm_ra_parts(a, wk, part, S, 4)
        with pytest.raises(cabi.LadeHipError, match="not in this build"):
            ops.gemm_parts(a, wk, part, S, 128, 4, 2, 2, ring=10)
        return
    want = torch.empty_like(part)
    ops.gemm_parts(a, wk, want, S, 128, 4)
    ops.gemm_ra_parts(a, wk, part, S, 4)
    assert torch.equal(part, want)
    for (bn, mt, nt) in ((128, 2, 2), (64, 2, 1), (128, 4, 1)):
        for ring in (10, 12, 14):
            ops.gemm_parts(a, wk, part, S, bn, 4, mt, nt, ring=ring)
            assert torch.allclose(part.view(S, M, N).sum(0), want.view(S, M, N).sum(0), atol=1e-3, rtol=1e-4), (bn, mt, nt, ring)


def test_16_row_granular_tiles_are_refused_by_the_default_build_and_correct_in_the_experimental_one():
    """csrc/gemm16.hpp (v_mfma_f32_16x16x32 tiles, 80 / 112 / 144 / 176-row activation tiles; probe state: split-K partials): measured 0-4 % against the
    padded 32-row classes and not integrated - the default build answers mt = 16 with an error code; the experimental build's partials sum to
    the fp64 product (another summation order than the 32x32x16 kernel's: not bit for bit)."""
    from lookaheaddecoding_amd import ops
    torch.manual_seed(13)
    for (M, N, K, S, bn, nt16) in ((76, 1024, 1024, 2, 128, 1), (138, 1536, 512, 2, 256, 2), (174, 1000, 768, 3, 128, 2), (100, 520, 384, 2, 128, 1)):
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        part = torch.full((S * M * N,), float("nan"), dtype=torch.float32, device="cuda")
        if not cabi.experimental():
            with pytest.raises(cabi.LadeHipError, match="not in this build"):
                ops.gemm_parts(a, ops.to_ktile(w), part, S, bn, (M + 15) // 16, 16, nt16)
            return
        for wt in (w, ops.to_ktile(w)):
            part.fill_(float("nan"))
            ops.gemm_parts(a, wt, part, S, bn, (M + 15) // 16, 16, nt16)
            got = part.view(S, M, N).double().sum(0)
            assert torch.allclose(got, a.double() @ w.double().t(), atol=2e-5, rtol=1e-5), (M, N, K, S, bn, nt16, wt.dim())
