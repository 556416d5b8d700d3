"""Tensor-level wrappers over the C ABI (one python function per HIP entry point).

Every function validates shapes/dtypes on the host, passes raw device pointers to liblade_hip.so
and returns torch tensors that live on the GPU.  Nothing here computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import cabi
from .cabi import AttnArgs, MaskParams, call, dtype_code, ptr


@dataclass
class StepMask:
    """Closed-form description of the lookahead mask of one step (see include/lade_hip.h).
    Reference: j_make_causal_mask_multilevel, lade/models/modeling_llama.py:115-207."""
    T: int
    P: int
    is_prefill: bool
    s: int = 1
    lguess: int = 0
    gs: int = 1
    level_offset: int = 0
    dist_offset: int = 0
    layout: int = 0          # 0: rows level-major (eager order); 1: levels >= 1 column-major (the reference's flash order)

    @staticmethod
    def from_levels(n_input: int, level_sizes, lguess: int, gs: int, P: int, is_prefill: bool = False, layout: int = 0) -> "StepMask":
        T = n_input + sum(level_sizes) + lguess
        return StepMask(T=T, P=P, is_prefill=is_prefill, s=level_sizes[-1], lguess=lguess, gs=gs, level_offset=n_input - 1,
                        dist_offset=1 + level_sizes[0] - level_sizes[-1], layout=layout)

    def c_struct(self) -> MaskParams:
        return MaskParams(self.T, self.P, int(self.is_prefill), self.s, self.lguess, self.gs, self.level_offset, self.dist_offset, self.layout)


def _dev(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise cabi.LadeHipError(f"{name} must be a GPU tensor (the HIP path has no CPU fallback)")


def attn_block_rows(n_rep: int, T: int) -> int:
    """query rows of one attention work-group (the kernel's shapes: 32 / 64 / 128 rows of the (head-in-group, token) space)"""
    rows = n_rep * T
    return 32 if rows <= 32 else (64 if rows <= 64 else 128)


def choose_splits(H: int, n_rep: int, T: int, S_tot: int, n_cu: int = 256, allow_single: bool = True, block_rows: int = 0, mode: int = 0) -> int:
    """KV splits of the lookahead attention.  One CU ingests only ~55-68 GB/s from HBM (tools/hbm_probe), so the
    grid (row blocks x KV heads x splits) should cover the CUs; a split is a contiguous range of 64-key tiles.
    block_rows: the work-group shape the launch will use (0: the 128-row default for more than 64 rows, else the smallest that holds them).
    mode 0: the sqrt rule below; 1: fill the CUs regardless of the merge cost; 2: half the sqrt rule's count (GQA heads whose partial
    round trip dominates) - the engine's in-step autotune picks (shape, mode) per row class (StepEngine._refine_in_step)."""
    br = block_rows or attn_block_rows(n_rep, T)
    blocks = (H // n_rep) * ((n_rep * T + br - 1) // br)
    tiles = max(1, (S_tot + 63) // 64)
    if tiles <= 5 and allow_single:               # <= 320 keys: one work-group per head beats a second (merge) launch
        return 1
    forced = int(cabi.debug("attn_split_rule", "0"))     # experiments: 1 = fill the CUs regardless of the merge cost
    fill = max(1, n_cu // max(blocks, 1))
    if forced == 1 or mode == 1:
        want = max(1, min(fill, tiles, 32))
    else:
        # streaming time per work-group falls as tiles/splits, the merge (and the partial round trip) grows with splits: the measured
        # optimum follows sqrt(tiles) - 5 splits at 18 tiles, 6 at 34, 8 at 65 (round 2, isolated pair at 34 tiles: 16.8 us with 6
        # splits, 17.1 us with 8)
        want = max(1, min(fill, math.isqrt(max(tiles - 1, 0)) + 1, tiles, 32))
        if mode == 2:
            want = max(1 if allow_single else 2, (want + 1) // 2)
    tps = (tiles + want - 1) // want              # tiles per split
    return (tiles + tps - 1) // tps               # drop the splits that would be empty


_HALVES_OK: dict = {}


def _check_rope_halves(cos: torch.Tensor, sin: torch.Tensor) -> None:
    """the fused RoPE loads ONE 16-byte vector per table for columns i.. and i + d/2..: valid for rotary tables built as cat(freqs, freqs)
    (lade/models/modeling_llama.py:252), which is checked once per table"""
    key = (cos.data_ptr(), sin.data_ptr(), tuple(cos.shape), cos._version, sin._version)
    if key not in _HALVES_OK:
        if torch.cuda.is_current_stream_capturing():
            return                                # (checked at the eager warm-up that precedes every capture)
        h = cos.shape[1] // 2
        if not (torch.equal(cos[:, :h], cos[:, h:]) and torch.equal(sin[:, :h], sin[:, h:])):
            raise cabi.LadeHipError("fused RoPE needs cos / sin tables whose two halves are equal (emb = cat(freqs, freqs))")
        if len(_HALVES_OK) > 256:
            _HALVES_OK.clear()
        _HALVES_OK[key] = True


def attn_fwd(q: Optional[torch.Tensor], k_cache: torch.Tensor, vt_cache: torch.Tensor, mask: StepMask, *, H: int, Hkv: int, d: int,
             out: Optional[torch.Tensor] = None, n_splits: Optional[int] = None, scale: Optional[float] = None,
             q_row_stride: Optional[int] = None, part_o: Optional[torch.Tensor] = None, part_ml: Optional[torch.Tensor] = None,
             dyn_P: Optional[torch.Tensor] = None, wg_rows: int = 0, qkv_parts: Optional[torch.Tensor] = None, n_parts: int = 0,
             positions: Optional[torch.Tensor] = None, cos: Optional[torch.Tensor] = None, sin: Optional[torch.Tensor] = None,
             sync_flags: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Lookahead attention for one step.  q: [T, >=H*d] rows (token stride q_row_stride elements, default
    q.stride(0)); k_cache [Hkv, S_max, d]; vt_cache [Hkv, d, S_max]; returns out [T, H*d].
    wg_rows: work-group shape (0 = default, 32 | 64 | 128 query rows per work-group).
    Fused RoPE + KV append (qkv_parts given): q is not read; q and the new K / V rows P..P+T come from the qkv projection's n_parts (1..4)
    fp32 split-K partials `qkv_parts` ([>= n_parts][T][(H + 2 Hkv) d] with stride qkv_parts.stride(0), or flat with stride T (H + 2 Hkv) d), rotated with the
    cos / sin rows positions[t] (positions None: row t) - the rows are written to the caches and the result equals
    rope_kv_append_parts + attn_fwd bit for bit.  sync_flags (device int32 [>= Hkv], zero before the first launch; needs q = a [T, H*d]
    buffer and n_splits > 1): the producer mode of the fused form - dedicated work-groups do the RoPE + append work once per KV head
    and hand it to the attention work-groups inside the launch (include/lade_hip.h)."""
    for n, t in (("k_cache", k_cache), ("vt_cache", vt_cache)):
        _dev(t, n)
    T = mask.T
    S_max = k_cache.shape[1]
    assert k_cache.shape == (Hkv, S_max, d) and vt_cache.shape == (Hkv, d, S_max), (k_cache.shape, vt_cache.shape)
    assert k_cache.is_contiguous() and vt_cache.is_contiguous()
    dt = k_cache.dtype
    fused = qkv_parts is not None
    if fused:
        assert n_parts >= 1 and cos is not None and sin is not None and qkv_parts.dtype == torch.float32 and dt != torch.float32
        assert cos.dtype == dt and cos.shape[1] == d and cos.is_contiguous() and sin.is_contiguous() and sin.shape == cos.shape
        assert positions is None or (positions.dtype == torch.int32 and positions.numel() >= T)
        row_w = (H + 2 * Hkv) * d
        part_stride = qkv_parts.stride(0) if qkv_parts.dim() == 3 else T * row_w
        assert qkv_parts.numel() >= (n_parts - 1) * part_stride + T * row_w
        _check_rope_halves(cos, sin)
        if sync_flags is not None:
            assert q is not None and q.dtype == dt and q.stride(-1) == 1 and sync_flags.dtype == torch.int32 and sync_flags.numel() >= Hkv
    else:
        _dev(q, "q")
        assert q.stride(-1) == 1 and q.dtype == dt and sync_flags is None
    dev = k_cache.device
    if out is None:
        out = torch.empty(T, H * d, dtype=dt, device=dev)
    if n_splits is None:
        n_splits = 1 if dt == torch.float32 else choose_splits(H, H // Hkv, T, mask.P + T)
    if dt == torch.float32:
        n_splits = 1
    if n_splits > 1:
        if part_o is None:
            part_o = torch.empty(n_splits, T, H, d, dtype=dt, device=dev)
        if part_ml is None:
            part_ml = torch.empty(n_splits, H, T, 2, dtype=torch.float32, device=dev)
    a = AttnArgs(ptr(q), ptr(k_cache), ptr(vt_cache), ptr(out), ptr(part_o), ptr(part_ml), ptr(dyn_P),
                 (q_row_stride if q_row_stride is not None else q.stride(0)) if q is not None else 0, out.stride(0), H, Hkv, d, S_max,
                 DTYPE_CODE_OF(dt), n_splits, scale if scale is not None else 1.0 / math.sqrt(d), mask.c_struct(), int(wg_rows))
    if fused:
        a.n_parts, a.qkv_parts, a.part_stride = int(n_parts), ptr(qkv_parts), int(part_stride)
        a.positions, a.cos_tab, a.sin_tab, a.max_pos = ptr(positions), ptr(cos), ptr(sin), int(cos.shape[0])
        a.sync_flags = ptr(sync_flags)
    call("lade_attn_fwd", C.byref(a))
    if n_splits > 1:
        call("lade_attn_combine", C.byref(a))
    return out


def DTYPE_CODE_OF(dt: torch.dtype) -> int:
    try:
        return cabi.DTYPE_CODE[dt]
    except KeyError:
        raise cabi.LadeHipError(f"unsupported dtype {dt}")


def time_attn(q, k_cache, vt_cache, mask: StepMask, *, H, Hkv, d, n_splits: int, reps: int = 20, debug_timeline: bool = False, wg_rows: int = 0):
    """Mean duration in microseconds of one attention launch (+combine), measured with hipEvents on the launch stream
    inside the library.  k_cache / vt_cache may be LISTS of caches: they are used round-robin, one per repetition, so
    that with enough of them (total > the 256 MB Infinity Cache) every launch streams its K/V from HBM like the
    consecutive layers of a decode step; a single cache stays Infinity-Cache resident across repetitions."""
    ks = list(k_cache) if isinstance(k_cache, (list, tuple)) else [k_cache]
    vs = list(vt_cache) if isinstance(vt_cache, (list, tuple)) else [vt_cache]
    assert len(ks) == len(vs)
    T, S_max = mask.T, ks[0].shape[1]
    out = torch.empty(T, H * d, dtype=q.dtype, device=q.device)
    part_o = torch.empty(max(n_splits, 1), T, H, d, dtype=q.dtype, device=q.device)
    n_ml = max(n_splits, 1) * H * T * 2
    part_ml = torch.zeros(n_ml + 16 * 4096, dtype=torch.float32, device=q.device)   # + room for LADE_DEBUG=attn_dbg=16 timestamps
    arr = (AttnArgs * len(ks))()
    for i, (k, v) in enumerate(zip(ks, vs)):
        arr[i] = AttnArgs(ptr(q), ptr(k), ptr(v), ptr(out), ptr(part_o), ptr(part_ml), None, q.stride(0), out.stride(0),
                          H, Hkv, d, S_max, dtype_code(q), n_splits, 1.0 / math.sqrt(d), mask.c_struct(), int(wg_rows))
    us = C.c_float(0.0)
    if len(ks) == 1:
        call("lade_time_attn", C.byref(arr[0]), reps, C.byref(us))
    else:
        call("lade_time_attn_rot", arr, len(ks), reps, C.byref(us))
    if debug_timeline:
        torch.cuda.synchronize()
        return float(us.value), part_ml[n_ml:].view(torch.int64).view(-1, 8).cpu()
    return float(us.value)


def mask_render(mask: StepMask, device="cuda") -> torch.Tensor:
    out = torch.zeros(mask.T, mask.P + mask.T, dtype=torch.uint8, device=device)
    m = mask.c_struct()
    call("lade_mask_render", C.byref(m), ptr(out))
    return out


def rope_kv_append(qkv: torch.Tensor, positions: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, k_cache: torch.Tensor,
                   vt_cache: torch.Tensor, P: int, *, H: int, Hkv: int, d: int, dyn_P: Optional[torch.Tensor] = None) -> None:
    """qkv [T, (H+2Hkv)*d] (q rotated in place); k_cache [Hkv,S_max,d]; vt_cache [Hkv,d,S_max]."""
    _dev(qkv, "qkv")
    T = qkv.shape[0]
    assert qkv.is_contiguous() and qkv.shape[1] == (H + 2 * Hkv) * d
    assert positions.dtype == torch.int32 and positions.numel() >= T
    assert cos.dtype == qkv.dtype and cos.shape[1] == d and cos.is_contiguous() and sin.is_contiguous()
    call("lade_rope_kv_append", ptr(qkv), ptr(positions), ptr(cos), ptr(sin), ptr(k_cache), ptr(vt_cache), T, P, ptr(dyn_P), H, Hkv,
         d, k_cache.shape[1], cos.shape[0], dtype_code(qkv))


def kv_commit(cache: torch.Tensor, src: int, dst: int, cnt: int, ctl: Optional[torch.Tensor] = None) -> None:
    """cache [L, 2, Hkv*S_max*d] viewed as K [Hkv,S_max,d] / V^T [Hkv,d,S_max] per layer."""
    L = cache.shape[0]
    meta = cache._lade_meta
    call("lade_kv_commit", ptr(cache), cache.stride(0), cache.stride(1), L, meta["Hkv"], meta["d"], meta["S_max"], src, dst, cnt,
         ptr(ctl), 0, cache.element_size())


def argmax_rows(logits: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev(logits, "logits")
    assert logits.dim() == 2 and logits.stride(1) == 1
    rows, V = logits.shape
    if out is None:
        out = torch.empty(rows, dtype=torch.int32, device=logits.device)
    call("lade_argmax_rows", ptr(logits), logits.stride(0), rows, V, dtype_code(logits), ptr(out))
    return out


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert x.is_contiguous() and x.dim() == 2
    if out is None:
        out = torch.empty_like(x)
    call("lade_rmsnorm", ptr(x), ptr(w), ptr(out), x.shape[0], x.shape[1], eps, dtype_code(x))
    return out


def embed_rmsnorm(table: torch.Tensor, ids: torch.Tensor, x: torch.Tensor, w: torch.Tensor, eps: float, out: torch.Tensor, rows: int) -> torch.Tensor:
    """x[r] = table[ids[r]]; out[r] = rmsnorm(x[r]) for r < rows: the step's embedding lookup and its first norm in one launch."""
    assert table.is_contiguous() and x.is_contiguous() and out.is_contiguous() and ids.dtype == torch.int32 and table.shape[1] == x.shape[1]
    call("lade_embed_rmsnorm", ptr(table), table.shape[0], ptr(ids), ptr(x), ptr(w), ptr(out), rows, x.shape[1], eps, dtype_code(x))
    return out


def add_rmsnorm(x: torch.Tensor, r: torch.Tensor, w: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x += r (in place); returns rmsnorm(x)."""
    assert x.is_contiguous() and r.is_contiguous() and x.shape == r.shape
    if out is None:
        out = torch.empty_like(x)
    call("lade_add_rmsnorm", ptr(x), ptr(r), ptr(w), ptr(out), x.shape[0], x.shape[1], eps, dtype_code(x))
    return out


def add_rmsnorm_rows(x: torch.Tensor, sel: torch.Tensor, n_sel: int, w: torch.Tensor, eps: float, *, r: Optional[torch.Tensor] = None,
                     part: Optional[torch.Tensor] = None, n_parts: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[j] = rmsnorm(x[sel[j]] + res[sel[j]]), res = r or the sum of the first n_parts fp32 split-K partials in `part`
    ([n_parts][rows][hidden]); x stays as it is.  The step's tail: only the rows whose logits are read."""
    rows, hidden = x.shape
    assert x.is_contiguous() and sel.dtype == torch.int32 and (r is None) != (part is None)
    if out is None:
        out = torch.empty(n_sel, hidden, dtype=x.dtype, device=x.device)
    call("lade_add_rmsnorm_rows", ptr(x), ptr(r), ptr(part), n_parts if part is not None else 0, rows * hidden, ptr(sel), ptr(w), ptr(out), n_sel, rows, hidden,
         eps, dtype_code(x))
    return out[:n_sel]


def silu_mul(gu: torch.Tensor, out: Optional[torch.Tensor] = None, layout: int = 0) -> torch.Tensor:
    """layout 0: gu rows = [gate | up]; 1: 16-row interleaved groups (the engine's fused gate/up order, see interleave_gate_up)"""
    assert gu.is_contiguous() and gu.shape[1] % 2 == 0
    inter = gu.shape[1] // 2
    if out is None:
        out = torch.empty(gu.shape[0], inter, dtype=gu.dtype, device=gu.device)
    call("lade_silu_mul", ptr(gu), ptr(out), gu.shape[0], inter, layout, dtype_code(gu))
    return out


def interleave_gate_up(wg: torch.Tensor, wu: torch.Tensor) -> torch.Tensor:
    """[inter, hid] x 2 -> fused [2*inter, hid] in groups of 16 rows: rows 32p .. 32p+15 = gate[16p ..], rows 32p+16 .. 32p+31 =
    up[16p ..].  In an MFMA 32x32 output tile a lane's accumulators e and e+8 are then gate and up of the same column."""
    inter, hid = wg.shape
    assert wu.shape == wg.shape and inter % 16 == 0
    return torch.stack([wg.view(inter // 16, 16, hid), wu.view(inter // 16, 16, hid)], dim=1).reshape(2 * inter, hid).contiguous()


def gather_rows(src: torch.Tensor, idx: torch.Tensor, out: Optional[torch.Tensor] = None, rows: Optional[int] = None) -> torch.Tensor:
    assert src.is_contiguous() and idx.dtype == torch.int32
    rows = idx.numel() if rows is None else rows
    if out is None:
        out = torch.empty(rows, src.shape[1], dtype=src.dtype, device=src.device)
    call("lade_gather_rows", ptr(src), ptr(idx), ptr(out), rows, src.shape[1], src.element_size(), src.shape[0])
    return out


def softmax_rows(logits: torch.Tensor, temperature: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    assert logits.dim() == 2 and logits.stride(1) == 1
    probs = out if out is not None else torch.empty(logits.shape, dtype=torch.float32, device=logits.device)
    assert probs.dtype == torch.float32 and probs.numel() >= logits.numel()
    call("lade_softmax_rows", ptr(logits), logits.stride(0), logits.shape[0], logits.shape[1], dtype_code(logits), float(temperature), ptr(probs))
    return probs


WARP_MAX_V = 1 << 24      # lade_warp_rows: a row of <= 32768 tokens lives in one work-group's registers, a larger one in the output row (L2)


def warp_rows(logits: torch.Tensor, rows: int, skip: int, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Temperature -> top-k -> top-p (HF warper semantics) in one launch.  logits [>= rows + skip, V]: logical row 0 is physical row 0,
    logical row r > 0 is physical row r + skip (out row + the candidate rows behind the window rows).  Returns fp32 [rows, V] with the
    removed tokens at -inf."""
    _dev(logits, "logits")
    assert logits.dim() == 2 and logits.stride(1) == 1
    V = logits.shape[1]
    assert logits.shape[0] >= (rows + skip if rows > 1 else 1)
    if out is None:
        out = torch.empty(rows, V, dtype=torch.float32, device=logits.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= rows * V
    call("lade_warp_rows", ptr(logits), logits.stride(0), rows, V, dtype_code(logits), float(temperature), int(top_k), float(top_p), int(skip), ptr(out))
    return out[:rows] if out.dim() == 2 else out


def softmax_gather(logits: torch.Tensor, rows: int, skip: int, guess: torch.Tensor, g: int, gs: int, g_cap: int, temperature: float,
                   scal: torch.Tensor, stats: torch.Tensor, g_dev: Optional[torch.Tensor] = None) -> None:
    """Device side of the sampling verify: per-row softmax statistics + the draft probabilities the acceptance loop reads
    (see lade_softmax_gather in include/lade_hip.h).  logits [>= rows + skip, V]; scal [rows, g_cap] fp32; stats [rows, 2] fp32."""
    assert logits.dim() == 2 and logits.stride(1) == 1 and scal.dtype == torch.float32 and stats.dtype == torch.float32
    V = logits.shape[1]
    assert logits.shape[0] >= (rows + skip if rows > 1 else 1) and scal.numel() >= rows * g_cap and stats.numel() >= 2 * rows
    call("lade_softmax_gather", ptr(logits), logits.stride(0), rows, V, dtype_code(logits), float(temperature), skip, ptr(guess), ptr(g_dev), g, gs,
         g_cap, ptr(scal), ptr(stats))


def to_ktile(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Row-major weight [N, K] -> the K-tile-major copy [K/64, N, 64] the skinny GEMM streams fastest (`lade_weight_to_ktile`): the
    128-byte segments of all N rows of one 64-deep K tile are contiguous.  The GEMM wrappers below tell the two layouts apart by the
    number of dimensions."""
    N, K = w.shape
    assert w.stride(1) == 1 and K % 64 == 0
    if out is None:
        out = torch.empty(K // 64, N, 64, dtype=w.dtype, device=w.device)
    assert out.shape == (K // 64, N, 64) and out.is_contiguous() and out.dtype == w.dtype
    call("lade_weight_to_ktile", ptr(w), w.stride(0), ptr(out), N, K, dtype_code(w))
    return out


def from_ktile(wkt: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """K-tile-major [K/64, N, 64] -> row-major [N, K] (`lade_weight_from_ktile`), into `out` when given (a reused scratch)"""
    KT, N, _ = wkt.shape
    assert wkt.is_contiguous() and wkt.shape[2] == 64
    if out is None:
        out = torch.empty(N, KT * 64, dtype=wkt.dtype, device=wkt.device)
    assert out.shape == (N, KT * 64) and out.stride(1) == 1 and out.dtype == wkt.dtype
    call("lade_weight_from_ktile", ptr(wkt), ptr(out), out.stride(0), N, KT * 64, dtype_code(wkt))
    return out


def _gemm(a: torch.Tensor, w: torch.Tensor, c, ldc: int, part, n_split: int, bn: int, mb: int, mt: int, nt: int, epilogue: int, ring: int = 0) -> None:
    M, K = a.shape
    assert a.stride(1) == 1
    if w.dim() == 3:                          # K-tile-major [K/64, N, 64]
        assert w.is_contiguous() and w.shape[0] * 64 == K and w.shape[2] == 64
        call("lade_gemm_skinny_kt", ptr(a), a.stride(0), ptr(w), c, ldc, part, M, w.shape[1], K, n_split, bn, mb, mt, nt, ring, epilogue, dtype_code(a))
    else:
        assert w.shape[1] == K and w.stride(1) == 1
        call("lade_gemm_skinny", ptr(a), a.stride(0), ptr(w), w.stride(0), c, ldc, part, M, w.shape[0], K, n_split, bn, mb, mt, nt, ring, epilogue, dtype_code(a))


def weight_rows(w: torch.Tensor) -> int:
    """N of a projection weight in either layout"""
    return w.shape[1] if w.dim() == 3 else w.shape[0]


def gemm_skinny(a: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor] = None, n_split: int = 1, bn: int = 128,
                part: Optional[torch.Tensor] = None, mb: int = 0, mt: int = 0, nt: int = 0, ring: int = 0) -> torch.Tensor:
    """out[M,N] = a[M,K] @ w[N,K]^T on the hand-written weight-streaming kernel (bf16 / f16); w row-major or K-tile-major (to_ktile).
    ring: stages of the LDS ring (0 = the shape's default)."""
    M, N = a.shape[0], weight_rows(w)
    if out is None:
        out = torch.empty(M, N, dtype=a.dtype, device=a.device)
    if n_split > 1 and part is None:
        part = torch.empty(n_split, M, N, dtype=torch.float32, device=a.device)
    _gemm(a, w, ptr(out), out.stride(0), ptr(part), n_split, bn, mb, mt, nt, 0, ring)
    if n_split > 1:
        call("lade_splitk_reduce", ptr(part), ptr(out), out.stride(0), M, N, n_split, dtype_code(a))
    return out


def gemm_argmax(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, pairs: Optional[torch.Tensor] = None, bn: int = 128, mb: int = 0, mt: int = 0,
                nt: int = 0, ring: int = 0) -> torch.Tensor:
    """out[m] = argmax_n (a @ w^T)[m][n] (int32, the value compared after rounding to the model dtype, first index wins ties) without
    materialising the product: the GEMM's epilogue leaves one (value, column) pair per row and column block in `pairs`
    (fp32 [M, ceil(N / bn), 2]), lade_argmax_pairs merges them.  Same ids as argmax_rows(gemm_skinny(a, w))."""
    M, N = a.shape[0], weight_rows(w)
    assert bn in (32, 64, 96, 128, 192, 224, 256) and M <= 128, "the pair buffer is indexed by the kernel's own column-block count"
    nb = (N + bn - 1) // bn
    if pairs is None:
        pairs = torch.empty(M * nb * 2, dtype=torch.float32, device=a.device)
    assert pairs.dtype == torch.float32 and pairs.numel() >= M * nb * 2 and out.dtype == torch.int32 and out.numel() >= M
    _gemm(a, w, None, 0, ptr(pairs), 1, bn, mb, mt, nt, 2, ring)
    call("lade_argmax_pairs", ptr(pairs), M, nb, ptr(out))
    return out


def gemm_swiglu(a: torch.Tensor, w_gu: torch.Tensor, out: torch.Tensor, bn: int, mb: int, mt: int = 0, nt: int = 0, ring: int = 0) -> torch.Tensor:
    """out[M, inter] = silu(a @ gate^T) * (a @ up^T) in ONE launch: w_gu is the 16-row interleaved fused weight (interleave_gate_up),
    row-major or K-tile-major; no split-K, no fp32 partials, no SwiGLU kernel."""
    assert out.shape == (a.shape[0], weight_rows(w_gu) // 2) and out.stride(1) == 1
    _gemm(a, w_gu, ptr(out), out.stride(0), None, 1, bn, mb, mt, nt, 1, ring)
    return out


def gemm_parts(a: torch.Tensor, w: torch.Tensor, part: torch.Tensor, n_split: int, bn: int, mb: int, mt: int = 0, nt: int = 0, ring: int = 0) -> None:
    """split-K GEMM that leaves its result as n_split fp32 partials in `part` ([n_split][M][N], contiguous) for a
    `*_parts` consumer kernel - no reduce pass."""
    M, N = a.shape[0], weight_rows(w)
    if n_split * M * N > part.numel():
        raise cabi.LadeHipError(f"split-K workspace too small: {n_split} x {M} x {N} fp32 partials > {part.numel()}")
    _gemm(a, w, None, 0, ptr(part), n_split, bn, mb, mt, nt, 0, ring)


RA_KT = 20          # K tiles a work-group of the register-resident-activation GEMM holds (gemm_decl.hpp)


def gemm_ra_parts(a: torch.Tensor, w_kt: torch.Tensor, part: torch.Tensor, n_split: int, cs: int = 4, n_groups: int = 0) -> None:
    """split-K GEMM with register-resident activations (M <= 128 rows, K-tile-major weight): n_split fp32 partials in `part`, bit-identical
    to gemm_parts with the same n_split."""
    M, K = a.shape
    assert a.stride(1) == 1 and w_kt.dim() == 3 and w_kt.is_contiguous() and w_kt.shape[0] * 64 == K and w_kt.shape[2] == 64
    N = w_kt.shape[1]
    if n_split * M * N > part.numel():
        raise cabi.LadeHipError(f"split-K workspace too small: {n_split} x {M} x {N} fp32 partials > {part.numel()}")
    call("lade_gemm_ra_kt", ptr(a), a.stride(0), ptr(w_kt), ptr(part), M, N, K, n_split, cs, n_groups, dtype_code(a))


def add_rmsnorm_parts(x: torch.Tensor, part: torch.Tensor, n_parts: int, w: torch.Tensor, eps: float, out: torch.Tensor) -> torch.Tensor:
    rows, hidden = x.shape
    call("lade_add_rmsnorm_parts", ptr(x), ptr(part), n_parts, rows * hidden, ptr(w), ptr(out), rows, hidden, eps, dtype_code(x))
    return out


def silu_mul_parts(part: torch.Tensor, n_parts: int, rows: int, inter: int, out: torch.Tensor, layout: int = 0) -> torch.Tensor:
    call("lade_silu_mul_parts", ptr(part), n_parts, rows * 2 * inter, ptr(out), rows, inter, layout, dtype_code(out))
    return out


def rope_kv_append_parts(part: torch.Tensor, n_parts: int, q_out: torch.Tensor, positions: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor,
                         k_cache: torch.Tensor, vt_cache: torch.Tensor, T: int, P: int, *, H: int, Hkv: int, d: int,
                         dyn_P: Optional[torch.Tensor] = None) -> None:
    call("lade_rope_kv_append_parts", ptr(part), n_parts, T * (H + 2 * Hkv) * d, ptr(q_out), ptr(positions), ptr(cos), ptr(sin), ptr(k_cache),
         ptr(vt_cache), T, P, ptr(dyn_P), H, Hkv, d, k_cache.shape[1], cos.shape[0], dtype_code(q_out))
