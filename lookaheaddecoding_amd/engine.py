"""Model step of lookahead decoding on MI355X: `jforward_multilevel` re-designed.

Reference: LlamaForCausalLM.jforward_multilevel -> LlamaModel.LlamaModeljforward ->
LlamaDecoderLayer.forward -> LlamaAttention.forward (lade/models/modeling_llama.py:1381-1608,
1108-1254, 822-899, 461-563).  There the step is ~100 small torch ops per layer, a dense fp32
mask, a torch.cat of the whole KV cache per layer and lm_head over all T rows.  Here one step is

    ids/positions (built on device)  -> embedding row gather
    per layer:  [add+]RMSNorm (HIP) -> fused QKV GEMM -> RoPE + in-place KV append (HIP)
                -> lookahead attention (HIP, mask in-kernel, split-KV + merge)
                -> O GEMM -> add+RMSNorm (HIP) -> fused gate/up GEMM (+ SwiGLU) -> down GEMM
    needed rows only -> add+RMSNorm -> lm_head GEMM -> row argmax (HIP, int32 ids)

with a preallocated KV cache ([L][2][Hkv*S_max*d]; keys row-major, values transposed) sized for
the 288 GB of HBM.  The projections of steps of <= 256 rows run on the hand-written weight-streaming
split-K GEMM (`lade_gemm_skinny_kt` on a K-tile-major copy of the weights where it fits the HBM, else
`lade_gemm_skinny` on the row-major ones; its fp32 partials are summed by the consumer kernels, the gate/up
GEMM carries SwiGLU in its epilogue) wherever the per-shape autotune finds it faster than the
library GEMM (hipBLASLt through torch.matmul); the lm_head runs on the same kernel without split-K (rows that are
read only); wider steps (prefill chunks) and fp32 use the library on the row-major weights.  No CPU fallback: construction fails without the HIP extension or without a GPU.
"""
from __future__ import annotations

import json
import math
import os
import sys
from typing import Dict, List, Optional, Sequence

import torch

from . import cabi, ops
from .ops import StepMask


def rope_inv_freq(d: int, theta: float, scaling: Optional[dict] = None) -> torch.Tensor:
    """inv_freq [d/2] in fp32.  Default: LlamaRotaryEmbedding (lade/models/modeling_llama.py:238-246).  `llama3`: the
    frequency-dependent rescaling newer Llama checkpoints carry in `rope_scaling` (low frequencies divided by `factor`,
    a smooth blend between `original_max_position_embeddings / low_freq_factor` and `/ high_freq_factor` wavelengths)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float32) / d))
    kind = None if not scaling else scaling.get("rope_type", scaling.get("type"))
    if kind == "llama3":
        factor = float(scaling["factor"])
        lo, hi = float(scaling.get("low_freq_factor", 1.0)), float(scaling.get("high_freq_factor", 4.0))
        old_len = float(scaling.get("original_max_position_embeddings", 8192))
        wavelen = 2 * math.pi / inv_freq
        scaled = torch.where(wavelen > old_len / lo, inv_freq / factor, inv_freq)
        smooth = (old_len / wavelen - lo) / (hi - lo)
        blended = (1 - smooth) * scaled / factor + smooth * scaled
        medium = ~(wavelen < old_len / hi) & ~(wavelen > old_len / lo)
        inv_freq = torch.where(medium, blended, scaled)
    elif kind not in (None, "default", "linear", "dynamic"):       # dynamic: the per-step rows come from StepEngine._ntk (lade_rope_rows_dynamic)
        raise cabi.LadeHipError(f"rope_scaling type {kind!r} is not implemented (default, linear, dynamic and llama3 are)")
    return inv_freq


def ntk_inv_freq_table(d: int, theta: float, factor: float, max_position_embeddings: int, S_max: int) -> torch.Tensor:
    """[n_len, d/2] fp32: row i = the inv_freq LlamaDynamicNTKScalingRotaryEmbedding._set_cos_sin_cache computes when it rebuilds at
    seq_len = max_position_embeddings + i (lade/models/modeling_llama.py:302-308; row 0 = the original base), with torch's own fp32 pow
    so that the values are the reference's."""
    mp = max_position_embeddings
    ar = torch.arange(0, d, 2).float() / d
    rows = [1.0 / (theta ** ar)]
    for L in range(mp + 1, max(S_max, mp) + 1):
        base = theta * ((factor * L / mp) - (factor - 1)) ** (d / (d - 2))
        rows.append(1.0 / (base ** ar))
    return torch.stack(rows).contiguous()


def rope_tables(d: int, max_pos: int, theta: float, dtype, device, scaling: Optional[dict] = None):
    """cos/sin [max_pos, d]: fp32 math, then cast - exactly LlamaRotaryEmbedding
    (lade/models/modeling_llama.py:238-256, cast at :264-265); `linear` scaling divides the positions by `factor`
    (LlamaLinearScalingRotaryEmbedding, :268-289)."""
    inv_freq = rope_inv_freq(d, theta, scaling)
    t = torch.arange(max_pos, dtype=torch.float32)
    if scaling and scaling.get("rope_type", scaling.get("type")) == "linear":
        t = t / float(scaling["factor"])
    freqs = torch.outer(t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype).to(device).contiguous(), emb.sin().to(dtype).to(device).contiguous()


def _on_device(fn):
    """Runs an engine / decoder method with the engine's GPU as the current device: the C-ABI calls launch on the CURRENT
    device's current stream, so a model on cuda:k driven from a thread whose current device is another GPU would otherwise
    launch cuda:k pointers on the wrong device."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        dev = self.device
        if torch.cuda.current_device() == dev.index:
            return fn(self, *a, **k)
        with torch.cuda.device(dev):
            return fn(self, *a, **k)
    return wrapped


from .tuning import (KernelDecisions, REP_ROWS, _STEP_TUNE_CACHE, _STEP_TUNE_LOCK, _TUNE_CACHE, _TUNE_LOCK, _TUNE_RANKED, _TUNE_TIMES,  # noqa: F401 (re-exported: tests / tools)
                     gpu_identity, shipped_tune_table)


class StepEngine(KernelDecisions):
    """Weights + KV cache + workspaces of one sequence (batch 1, as the reference asserts at
    lade/models/modeling_llama.py:1448)."""

    def __init__(self, cfg: dict, weights: Dict[str, torch.Tensor], *, dtype=torch.bfloat16, device="cuda", max_seq: int = 4096,
                 max_T: int = 512, consume_weights: bool = False):
        cabi.load_library()                      # fail loudly when the HIP extension is missing
        if not torch.cuda.is_available():
            raise cabi.LadeHipError("StepEngine needs a GPU (MI355X); the HIP path has no CPU fallback")
        self.cfg = dict(cfg)
        self.dtype = dtype
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.hidden, self.inter = cfg["hidden"], cfg["inter"]
        self.L, self.H, self.Hkv, self.d, self.V = cfg["layers"], cfg["heads"], cfg["kv_heads"], cfg["head_dim"], cfg["vocab"]
        self.eps = float(cfg["eps"])
        self.S_max = ((max_seq + 63) // 64) * 64
        self.generation = 0                  # bumped whenever buffers are re-allocated (captured graphs become stale)
        dev, dt = self.device, dtype

        self._aliased = set()               # (layer, projection) whose row-major weight is the CALLER's storage (HF drop-in: wo / wd)

        def W(k, tag=None):
            # consume_weights: the entry is removed from the caller's dict as soon as it has been fused into the engine's
            # own layout, so a 140 GB model (Llama-2-70B in bf16) is never held twice
            src = weights.pop(k) if consume_weights else weights[k]
            out = src.to(device=dev, dtype=dt)
            if tag is not None and not consume_weights and out.is_contiguous() and out.data_ptr() == src.data_ptr():
                self._aliased.add(tag)      # `.to()` was a no-op: releasing the engine's reference would free nothing
            return out

        # tied embeddings: same storage (HF hands out a fresh tensor wrapper per `.weight.data`, so identity of the python objects
        # says nothing)
        lm, em = weights["lm_head"], weights["embed"]
        tied = lm is em or (lm.shape == em.shape and lm.dtype == em.dtype and lm.device == em.device and lm.data_ptr() == em.data_ptr())
        self.embed = W("embed").contiguous()
        self.norm_w = W("norm").contiguous()
        self._lm_head, self._lm_kt = (self.embed if tied else W("lm_head").contiguous()), None
        if tied and consume_weights:
            weights.pop("lm_head", None)
        self.layers: List[dict] = []
        for i in range(self.L):
            p = f"layers.{i}."
            self.layers.append(dict(
                ln1=W(p + "ln1").contiguous(), ln2=W(p + "ln2").contiguous(),
                wqkv=torch.cat([W(p + "wq"), W(p + "wk"), W(p + "wv")], dim=0).contiguous(),
                wo=W(p + "wo", (i, "wo")).contiguous(),
                wgu=self._fuse_gate_up(W(p + "wg"), W(p + "wu")),
                wd=W(p + "wd", (i, "wd")).contiguous()))
        # hand-written weight-streaming GEMM (split-K partials consumed by the fused glue kernels) for steps of
        # <= 256 tokens; every (N, K, row class) is timed against the library GEMM once and the faster one is kept
        self.custom_gemm = dt != torch.float32 and os.environ.get("LADE_GEMM", "1") != "0" and self.d % 16 == 0 and self.hidden % 64 == 0 and self.inter % 64 == 0
        self.ktile, self.ktile_only, self.ktile_bytes = False, False, 0
        self._build_ktile_copies()
        self._alloc_cache(self.S_max)
        self.attn_events = None             # set to a list to collect (start, end, T, n_splits) hipEvent pairs per layer
        self.attn_events_empty = None       # ... and back-to-back event pairs (what an empty bracket reads)
        self.skip_attn = False              # bench.py only: True = leave the attention launches out, RoPE + KV append as a launch of their own (step-time
                                            # difference = what attention costs on top of them); "all" = leave out RoPE + append as well
        self.max_splits = 32
        self.gemm_cfg = {}
        self.gemm_times = {}                # (projection, row class) -> (ms of the chosen kernel in the autotune, weight bytes)
        self._ranked = {}                   # (projection, row class) -> the isolated pass's candidates, fastest first [(ms, cfg | None)]
        self._refined = set()               # row classes whose decisions were re-taken inside the step (_refine_in_step)
        self._refining = False
        # attention launch parameters per row class: (RoPE + KV append fused into the launch, work-group rows, split mode of ops.choose_splits);
        # the default until (unless) the in-step pass decides: RoPE + append as a launch of their own (the fused form measured +3.7 us per
        # layer at the 7B step - every split re-reads the q rows as fp32 partials; LADE_DEBUG=fuse_rope=1 makes it the default, the
        # in-step pass always tries both), the 128-row shape, the sqrt split rule
        self.attn_cfg = {}
        fuse_env = cabi.debug("fuse_rope", "0")
        if fuse_env in ("1", "2") and not cabi.experimental():
            raise cabi.LadeHipError("LADE_DEBUG=fuse_rope needs a library built with `make -C lookaheaddecoding_amd/csrc EXPERIMENTAL=1` (the fused forms measured slower: DESIGN 4.9)")
        # work-group rows of the attention launch: 128, but 64 where eight or more heads share a KV head (Llama-2-70B: 480 rows per KV head at T = 60) -
        # 64 row blocks per launch already fill a quarter of the CUs, the split rule then takes 4 splits instead of 6: the same time in the step (config 5:
        # 24.85 / 24.80 vs 24.84 / 24.79 ms alternating) for 1.85 x instead of 2.25 x the algorithmic HBM bytes (profiles/r6_attn_gqa_sweep.txt, r6_c5_gqa_launch_ab.txt)
        self.attn_default = (int(fuse_env) if fuse_env in ("1", "2") else 0, 128, 0)
        if self.H // self.Hkv >= 8:
            for mclass in self.ROW_CLASSES:
                if (self.H // self.Hkv) * (mclass - 31) > 128:          # every step of the class stacks more than 128 rows on a KV head
                    self.attn_cfg[mclass] = (self.attn_default[0], 64, 0)
        self.step_tune_log = {}             # row class -> what the in-step pass measured (bench.py prints it)
        self._alloc_workspaces(max_T)
        try:
            self.n_cu = torch.cuda.get_device_properties(self.device.index).multi_processor_count
        except AssertionError:      # device count not initialised on this thread yet
            self.n_cu = 256
        # persisted kernel decisions: LADE_TUNE_FILE=<json> (read, and extended when this process tunes), else the table SHIPPED with the package
        # for this GPU model (lookaheaddecoding_amd/tuned/<isa>_<CUs>cu.json: the BASELINE model shapes, so that every MI355X launches the same
        # kernels for them - one 16-bit token stream per model, no multi-second tuning before the first request); LADE_TUNE_FILE=off: neither
        env_tf = os.environ.get("LADE_TUNE_FILE") or None
        self.tune_file = None if env_tf in (None, "off", "none", "0") else env_tf
        self.tune_loaded = []               # row classes whose decisions came from a tune file
        self.tune_source = None
        if self.custom_gemm and env_tf not in ("off", "none", "0"):
            if self.tune_file:
                self._load_tune_file(self.tune_file, strict=True)
            if not self.tune_loaded:
                shipped = shipped_tune_table(self.device)
                if os.path.exists(shipped):
                    self._load_tune_file(shipped, strict=False)

    def _fuse_gate_up(self, wg: torch.Tensor, wu: torch.Tensor) -> torch.Tensor:
        """gate and up projections as ONE weight.  16-row interleaved ([16 gate rows | their 16 up rows] per 32-row MFMA tile) so that
        the GEMM's SwiGLU epilogue is lane-local; every consumer of the fused output (lade_silu_mul*, layout 1) knows the order.
        Shapes whose intermediate size is not a multiple of 16 keep the plain [gate | up] concatenation (layout 0)."""
        self.gu_layout = 1 if wg.shape[0] % 16 == 0 else 0
        return ops.interleave_gate_up(wg, wu) if self.gu_layout else torch.cat([wg, wu], dim=0).contiguous()

    # ---- weight layout of the decode GEMMs ----------------------------------------------------------------
    KTILE_RESERVE = 24 << 30             # HBM left free for the KV cache, workspaces and the library's own buffers when the copies are made

    def _build_ktile_copies(self) -> None:
        """K-TILE-MAJOR projection weights ([K/64][N][64], `lade_weight_to_ktile`) for the skinny GEMM: the 128-byte segments of all rows of
        one K tile are contiguous, so a work-group's tile is one contiguous read and the work-groups of a split sweep memory linearly
        (7B projections at 60 rows: 88.8 -> 79.2 us per layer, 70B: 296.5 -> 261.4, bit-identical results).  The library
        GEMM (prefill chunks where it is the faster kernel) reads row-major weights, so the HBM decides what is held:
          * "dual" (everything fits twice: 13 GB more at 7B, 26 GB at 13B of the 288 GB): a second copy of every projection;
          * otherwise (Llama-2-70B in bf16) every projection the engine OWNS is converted layer by layer and its row-major original is
            kept only while the HBM budget lasts (free memory - KV cache - workspaces - KTILE_RESERVE): the first layers stay in both
            layouts, the rest K-tile-major only - their library GEMMs get a row-major operand from `lade_weight_from_ktile` into one
            scratch per projection.  (Round 3 released every original: 70B prefill paid 320 rebuilds, 7.1 k -> 6.1 k tokens/s, while
            ~120 GB of HBM sat unused.)
          * a projection whose row-major weight is the CALLER's storage (HF drop-in without consume_weights: o / down are the module's
            own tensors, `.to()` is a no-op) cannot be released by the engine: when the copies do not all fit it keeps the row-major
            layout for every kernel and gets NO K-tile copy (a copy would be pure extra memory).
        LADE_W_KTILE = 0 | 1 (dual) | only (release every owned original) overrides the decision; LADE_KTILE_BUDGET_MB the budget."""
        want = os.environ.get("LADE_W_KTILE", "auto")
        self.kt_names: tuple = ()            # projections streamed K-tile-major by the decode GEMMs
        self.rows_kept, self.rows_total = 0, 0
        if not self.custom_gemm or want == "0":
            return
        nbytes = lambda t: t.numel() * t.element_size()
        extra = sum(nbytes(lw[n]) for lw in self.layers for n in self.LAYER_GEMMS)
        torch.cuda.empty_cache()             # blocks sitting free in torch's caching allocator are memory the copies can use
        free, _total = torch.cuda.mem_get_info(self.device)
        # what is allocated after this: the KV cache and the step workspaces of the configured sizes, + a fixed reserve for the
        # library's workspaces, the n-gram pool and growth
        esz = self.layers[0]["wo"].element_size()
        later = 2 * self.L * self.Hkv * self.S_max * self.d * esz + 16 * 128 * max((self.H + 2 * self.Hkv) * self.d, 2 * self.inter) * 4
        budget = free - later - self.KTILE_RESERVE
        if os.environ.get("LADE_KTILE_BUDGET_MB"):
            budget = int(float(os.environ["LADE_KTILE_BUDGET_MB"]) * (1 << 20))
        dual = want == "1" or (want == "auto" and extra <= budget)
        aliased_names = {n for (_li, n) in self._aliased}
        names = self.LAYER_GEMMS if dual else tuple(n for n in self.LAYER_GEMMS if n not in aliased_names)
        with torch.cuda.device(self.device):            # the C ABI launches on the current device's current stream
            if not dual and names:
                # one scratch per projection for the library GEMM's row-major operand, allocated before the conversion frees anything
                self._row_scratch = {n: torch.empty_like(self.layers[0][n]) for n in names}
                budget -= sum(nbytes(t) for t in self._row_scratch.values())
            for lw in self.layers:
                for n in names:
                    lw[n + "_kt"] = ops.to_ktile(lw[n])
                    self.rows_total += 1
                    budget -= nbytes(lw[n])
                    if not dual and (want == "only" or budget < 0):
                        budget += nbytes(lw[n])
                        del lw[n]                       # the allocator hands the block to the next conversion
                    else:
                        self.rows_kept += 1
        self.kt_names = tuple(names)
        self.ktile = bool(names)
        self.ktile_only = bool(names) and self.rows_kept == 0          # no row-major original of a converted projection is left
        self.ktile_bytes = sum(nbytes(lw[n + "_kt"]) for lw in self.layers for n in names if n in lw) + nbytes(self._lm_head)
        self.lm_head = self._lm_head                 # builds its copy (the output projection is small: always held in both layouts)

    def refresh_ktile(self) -> None:
        """Re-derives the K-tile-major copies from the row-major weights after an IN-PLACE update of the latter (weight reload, adapter
        merge, an edit of tied embeddings): prefill and wide steps read the row-major tensors live - for o / down on the HF path those are
        the module's own storage - while decode steps stream the copies made at construction.  Projections whose row-major original was
        released have no second layout to go stale (edit `layers[i][name + "_kt"]` through `ops.to_ktile`).  Captured hipGraphs read the
        same buffers, so they stay valid."""
        with torch.cuda.device(self.device):
            for lw in self.layers:
                for n in self.kt_names:
                    if n in lw:
                        ops.to_ktile(lw[n], out=lw[n + "_kt"])
            self.lm_head = self._lm_head

    @property
    def lm_head(self) -> torch.Tensor:
        return self._lm_head

    @lm_head.setter
    def lm_head(self, w: torch.Tensor) -> None:
        """Replaces the output projection (bench / tests build successor-map and tied models on a live engine).  The K-tile-major copy
        the step's lm_head GEMM streams is rebuilt here; in-place edits of `w` made AFTER the assignment are not seen by the copy -
        assign again."""
        self._lm_head = w
        self._lm_kt = None
        if self.ktile and w.dim() == 2 and w.stride(1) == 1 and w.shape[1] % 64 == 0:
            with torch.cuda.device(self.device):
                self._lm_kt = ops.to_ktile(w)

    def zero_projections(self, names: Sequence[str]) -> None:
        """bench / tests (successor-map models): zero the named projections of every layer in every layout the engine holds"""
        for lw in self.layers:
            for n in names:
                for k in (n, n + "_kt"):
                    if k in lw:
                        lw[k].zero_()

    def _w(self, lw: dict, name: str) -> torch.Tensor:
        """the weight the skinny GEMM streams: the K-tile-major copy when the engine holds one"""
        kt = lw.get(name + "_kt")
        return kt if kt is not None else lw[name]

    def _row(self, lw: dict, name: str) -> torch.Tensor:
        """the row-major weight a library GEMM takes; K-tile-only engines rebuild it into the projection's scratch (same stream: ordered
        before the GEMM that reads it and after the previous layer's)"""
        w = lw.get(name)
        return w if w is not None else ops.from_ktile(lw[name + "_kt"], out=self._row_scratch[name])

    @staticmethod
    def _nk(w: torch.Tensor):
        """(N, K) of a projection weight in either layout"""
        return (w.shape[1], w.shape[0] * 64) if w.dim() == 3 else (w.shape[0], w.shape[1])

    def _alloc_cache(self, S_max: int, keep_rows: int = 0) -> None:
        """KV cache [L][2][Hkv*S_max*d] (K: [Hkv][S_max][d], V: [Hkv][d][S_max]), zero-initialised; RoPE tables for it.
        keep_rows > 0: the first keep_rows rows of the previous cache are carried over (growth under a live sequence)."""
        dev, dt = self.device, self.dtype
        old = getattr(self, "kv", None)
        old_k, old_v = getattr(self, "_k_views", None), getattr(self, "_vt_views", None)
        self.S_max = S_max
        self.cos, self.sin = rope_tables(self.d, max(self.cfg.get("max_pos", 4096), S_max), self.cfg.get("rope_theta", 10000.0), dt, dev,
                                         self.cfg.get("rope_scaling"))
        # rope_scaling "dynamic" (LlamaDynamicNTKScalingRotaryEmbedding, lade/models/modeling_llama.py:292-318): the tables above are never
        # read; every step gets the cos / sin rows of its own positions from the device-resident "longest length seen" (lade_rope_rows_dynamic)
        sc = self.cfg.get("rope_scaling") or {}
        prev = getattr(self, "_ntk", None)
        self._ntk = None
        if sc.get("rope_type", sc.get("type")) == "dynamic":
            mp = int(self.cfg.get("max_pos", 4096))
            tab = ntk_inv_freq_table(self.d, self.cfg.get("rope_theta", 10000.0), float(sc["factor"]), mp, S_max).to(dev)
            state = prev["state"] if prev is not None else torch.full((1,), mp, dtype=torch.int32, device=dev)
            self._ntk = dict(mp=mp, inv_tab=tab, state=state, rows=None)
        self.kv = torch.zeros(self.L, 2, self.Hkv * S_max * self.d, dtype=dt, device=dev)
        self.kv._lade_meta = dict(Hkv=self.Hkv, d=self.d, S_max=S_max)
        # per-layer views built once: an eager step issues ~10 launches per layer and must not spend its time in tensor indexing
        self._k_views = [self.kv[li, 0].view(self.Hkv, S_max, self.d) for li in range(self.L)]
        self._vt_views = [self.kv[li, 1].view(self.Hkv, self.d, S_max) for li in range(self.L)]
        if old is not None and keep_rows > 0:
            for li in range(self.L):
                self._k_views[li][:, :keep_rows].copy_(old_k[li][:, :keep_rows])
                self._vt_views[li][:, :, :keep_rows].copy_(old_v[li][:, :, :keep_rows])

    def _alloc_workspaces(self, max_T: int) -> None:
        """fixed addresses: graph-capturable, no allocator traffic in the loop"""
        dev, dt = self.device, self.dtype
        self.max_T = max_T
        qkv_w = (self.H + 2 * self.Hkv) * self.d
        self.ws_x = torch.empty(max_T, self.hidden, dtype=dt, device=dev)
        self.ws_h = torch.empty(max_T, self.hidden, dtype=dt, device=dev)
        self.ws_r = torch.empty(max_T, self.hidden, dtype=dt, device=dev)
        self.ws_qkv = torch.empty(max_T, qkv_w, dtype=dt, device=dev)
        self.ws_o = torch.empty(max_T, self.H * self.d, dtype=dt, device=dev)
        self.ws_gu = torch.empty(max_T, 2 * self.inter, dtype=dt, device=dev)
        self.ws_a = torch.empty(max_T, self.inter, dtype=dt, device=dev)
        if dt != torch.float32:
            self.part_o = torch.empty(self.max_splits * self.H * max_T * self.d, dtype=dt, device=dev)
            self.part_ml = torch.empty(self.max_splits * self.H * max_T * 2, dtype=torch.float32, device=dev)
        else:
            self.part_o = self.part_ml = None
        if self.custom_gemm:
            self.ws_part = torch.empty(16 * 128 * max(qkv_w, 2 * self.inter, self.hidden), dtype=torch.float32, device=dev)
            self.ws_q = torch.empty(max_T, self.H * self.d, dtype=dt, device=dev)
            if getattr(self, "attn_flags", None) is None:          # producer mode of the fused RoPE: one arrival counter per KV head, zero between launches
                self.attn_flags = torch.zeros(max(64, self.Hkv), dtype=torch.int32, device=dev)

    def grow(self, max_seq: int, max_T: int, keep_rows: int = 0) -> None:
        """Enlarges the KV cache and / or the step workspaces in place (the fused weights stay): the first keep_rows cache
        rows survive.  Captured hipGraphs over the old buffers are the caller's to drop (LookaheadDecoder re-captures when
        `generation` changes)."""
        S_new = ((max_seq + 63) // 64) * 64
        if S_new > self.S_max:
            self._alloc_cache(S_new, keep_rows)
            self.generation += 1
        if max_T > self.max_T:
            self._alloc_workspaces(max_T)
            self.generation += 1

    # ---- views --------------------------------------------------------------------------------
    def k_cache(self, layer: int) -> torch.Tensor:
        return self._k_views[layer]

    def vt_cache(self, layer: int) -> torch.Tensor:
        return self._vt_views[layer]

    def reset(self) -> None:
        """A new sequence on this model.  The dynamic-NTK "longest length seen" is NOT reset: in the reference it lives on the rotary module
        (`max_seq_len_cached` + the rebuilt inv_freq, lade/models/modeling_llama.py:243-246, :299-316) and survives from one generate() call to the
        next - a second, shorter generation rotates with the largest base the first one reached (tests/golden/e2e_dynamic_ntk_again.json:
        reference runs whose second call differs from a fresh model's).  `reset_rope_state()` is the fresh model."""
        self.kv.zero_()

    def reset_rope_state(self) -> None:
        """dynamic NTK only: forget the longest length seen (= a freshly constructed reference model: max_seq_len_cached = max_position_embeddings)"""
        if self._ntk is not None:
            self._ntk["state"].fill_(self._ntk["mp"])

    @property
    def ntk_state(self) -> Optional[torch.Tensor]:
        """device int32[1]: the longest kv_seq_len the dynamic-NTK rotary embedding has seen (None without `rope_scaling: dynamic`) - part
        of the sequence's state: whoever runs a throw-away forward (graph warm-up, tuning probe) saves and restores it"""
        return None if self._ntk is None else self._ntk["state"]

    def _ntk_rows(self, pos: torch.Tensor, T: int, P: int, dyn_P, rope_len: int, pad=None):
        """cos / sin rows of this step's T positions under dynamic NTK scaling; returns (row index 0..T-1, cos_rows, sin_rows) for the rope kernels.
        pad = (g_dev, gcap, gs): a hipGraph step padded to gcap candidates counts only its real rows (T - (gcap - g) gs), as the reference's
        kv_seq_len does."""
        nt = self._ntk
        if nt["rows"] is None or nt["rows"][0].shape[0] < self.max_T:
            nt["rows"] = (torch.empty(self.max_T, self.d, dtype=self.dtype, device=self.device), torch.empty(self.max_T, self.d, dtype=self.dtype, device=self.device),
                          torch.arange(self.max_T, dtype=torch.int32, device=self.device))
        cos_r, sin_r, iota = nt["rows"]
        g_dev, gcap, gs = pad if pad is not None else (None, 0, 1)
        cabi.call("lade_rope_rows_dynamic", cabi.ptr(pos), T, P, cabi.ptr(dyn_P), int(rope_len), cabi.ptr(nt["state"]), nt["mp"], cabi.ptr(nt["inv_tab"]),
                  nt["inv_tab"].shape[0], self.d, cabi.ptr(cos_r), cabi.ptr(sin_r), cabi.dtype_code(cos_r), cabi.ptr(g_dev), gcap, gs)
        return iota, cos_r, sin_r

    # ---- one forward -----------------------------------------------------------------------------
    @_on_device
    def forward(self, ids: torch.Tensor, pos: torch.Tensor, mask: StepMask, sel_rows: torch.Tensor, n_sel: int,
                dyn_P: Optional[torch.Tensor] = None, n_splits: Optional[int] = None, rope_len: int = 0, ntk_pad=None,
                argmax_out: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        """ids/pos: device int32 [>=T]; mask describes the step; sel_rows: device int32 [n_sel] rows whose
        logits are needed.  Appends the T new K/V rows at P..P+T and returns logits [n_sel, V] (model dtype,
        as `self.lm_head(hidden_states)` does at lade/models/modeling_llama.py:1541).
        argmax_out (device int32 [>= n_sel], greedy steps): the row argmax of those logits is written there and None is returned - on
        the hand-written lm_head GEMM the argmax runs in its epilogue and the logits are never materialised."""
        T, P = mask.T, mask.P                     # with dyn_P the kernels read P from the device (mask.P is then 0)
        if T > self.max_T or P + T > self.S_max:
            raise cabi.LadeHipError(f"step of T={T} tokens at P={P} exceeds the engine limits (max_T={self.max_T}, S_max={self.S_max})")
        H, Hkv, d = self.H, self.Hkv, self.d
        x, h, r = self.ws_x[:T], self.ws_h[:T], self.ws_r[:T]
        qkv, o, gu, a = self.ws_qkv[:T], self.ws_o[:T], self.ws_gu[:T], self.ws_a[:T]
        fused = self.custom_gemm and T <= self.ROW_CLASSES[-1]
        if fused and not self._refining:
            mclass = next(c for c in self.ROW_CLASSES if T <= c)
            if mclass not in self._refined:          # once per row class: the decisions re-taken INSIDE a step - BEFORE this call touches a workspace
                for n in self.LAYER_GEMMS:
                    self._tune(n, mclass)
                self._refine_in_step(mclass)         # (raises during a stream capture: a graph must never bake a table that later eager steps do not use)
                x, h, r = self.ws_x[:T], self.ws_h[:T], self.ws_r[:T]            # (the probe may have grown the workspaces)
                qkv, o, gu, a = self.ws_qkv[:T], self.ws_o[:T], self.ws_gu[:T], self.ws_a[:T]
        cfg_qkv = self._tune("wqkv", T) if fused else None
        cfg_o = self._tune("wo", T) if fused else None
        cfg_gu = self._tune("wgu", T) if fused else None
        cfg_d = self._tune("wd", T) if fused else None
        # the attention launch of this row class: RoPE + KV append inside it (from the qkv GEMM's partials) or as a launch of their own,
        # the work-group shape, the split rule
        acfg = (self.attn_cfg.get(next(c for c in self.ROW_CLASSES if T <= c), self.attn_default) if fused else (0, 0, 0))
        if n_splits is None:
            n_splits = self.n_splits_for(T, P + T, choice=acfg if fused else None)
        fuse_rope = bool(acfg[0]) and cfg_qkv is not None and cfg_qkv[2] <= 4 and not self.skip_attn
        fuse_embed = cabi.debug("fuse_tail", "1") != "0"      # embedding lookup inside the first layer's input norm (one launch less)
        if not (fuse_embed and self.layers):
            ops.gather_rows(self.embed, ids, out=x, rows=T)
        rpos, rcos, rsin = (pos, self.cos, self.sin) if self._ntk is None else self._ntk_rows(pos, T, P, dyn_P, rope_len, ntk_pad)
        part = self.ws_part if fused else None
        r_parts = 0                     # > 0: the pending residual branch lives in `part` as that many split-K partials
        for li, lw in enumerate(self.layers):
            if li == 0 and fuse_embed:
                ops.embed_rmsnorm(self.embed, ids, x, lw["ln1"], self.eps, h, T)
            elif li == 0:
                ops.rmsnorm(x, lw["ln1"], self.eps, out=h)
            elif r_parts:
                ops.add_rmsnorm_parts(x, part, r_parts, lw["ln1"], self.eps, out=h)   # x += mlp(prev); h = norm(x)
            else:
                ops.add_rmsnorm(x, r, lw["ln1"], self.eps, out=h)
            if cfg_qkv:
                ops.gemm_parts(h, self._w(lw, "wqkv"), part, cfg_qkv[2], cfg_qkv[1], cfg_qkv[0], cfg_qkv[3], cfg_qkv[4], cfg_qkv[5])
                qb = self.ws_q[:T]
                if not fuse_rope and self.skip_attn != "all":
                    ops.rope_kv_append_parts(part, cfg_qkv[2], qb, rpos, rcos, rsin, self.k_cache(li), self.vt_cache(li), T, P, H=H, Hkv=Hkv, d=d, dyn_P=dyn_P)
                q_in = qb
            else:
                torch.matmul(h, self._row(lw, "wqkv").t(), out=qkv)
                ops.rope_kv_append(qkv, rpos, rcos, rsin, self.k_cache(li), self.vt_cache(li), P, H=H, Hkv=Hkv, d=d, dyn_P=dyn_P)
                q_in = qkv
            ev = None if self.skip_attn else self.attn_events      # skip_attn: `o` keeps stale values, the logits are meaningless
            if ev is not None:              # bench.py: hipEvents around the attention launch pair, in the real step
                if self.attn_events_empty is not None:      # calibration: what two events recorded back to back read here
                    c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    c0.record()
                    c1.record()
                    self.attn_events_empty.append((c0, c1))
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            if fuse_rope:
                # q and the new K / V rows straight from the qkv GEMM's partials: no RoPE launch (same bits as the two-launch form).  Mode 2:
                # dedicated work-groups of the launch produce them once per KV head and hand them over through sync_flags
                prod = acfg[0] == 2 and n_splits > 1
                if prod:
                    self.attn_flags.zero_()          # (not left to the previous launch's combine: a launch that failed would leave stale arrival counts behind)
                ops.attn_fwd(qb if prod else None, self.k_cache(li), self.vt_cache(li), mask, H=H, Hkv=Hkv, d=d, out=o, n_splits=n_splits, part_o=self.part_o,
                             part_ml=self.part_ml, dyn_P=dyn_P, wg_rows=acfg[1], qkv_parts=part, n_parts=cfg_qkv[2], positions=rpos, cos=rcos, sin=rsin,
                             sync_flags=self.attn_flags if prod else None)
            elif not self.skip_attn:
                ops.attn_fwd(q_in, self.k_cache(li), self.vt_cache(li), mask, H=H, Hkv=Hkv, d=d, out=o, n_splits=n_splits,
                             part_o=self.part_o, part_ml=self.part_ml, dyn_P=dyn_P, wg_rows=acfg[1])
            if ev is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                ev.append((e0, e1, mask.T, n_splits))
            if cfg_o:
                ops.gemm_parts(o, self._w(lw, "wo"), part, cfg_o[2], cfg_o[1], cfg_o[0], cfg_o[3], cfg_o[4], cfg_o[5])
                ops.add_rmsnorm_parts(x, part, cfg_o[2], lw["ln2"], self.eps, out=h)      # x += attn; h = norm(x)
            else:
                torch.matmul(o, self._row(lw, "wo").t(), out=r)
                ops.add_rmsnorm(x, r, lw["ln2"], self.eps, out=h)
            if cfg_gu and cfg_gu[2] == 1:                 # gate/up GEMM + SwiGLU in one launch
                ops.gemm_swiglu(h, self._w(lw, "wgu"), a, cfg_gu[1], cfg_gu[0], cfg_gu[3], cfg_gu[4], cfg_gu[5])
            elif cfg_gu:
                ops.gemm_parts(h, self._w(lw, "wgu"), part, cfg_gu[2], cfg_gu[1], cfg_gu[0], cfg_gu[3], cfg_gu[4], cfg_gu[5])
                ops.silu_mul_parts(part, cfg_gu[2], T, self.inter, out=a, layout=self.gu_layout)
            else:
                torch.matmul(h, self._row(lw, "wgu").t(), out=gu)
                ops.silu_mul(gu, out=a, layout=self.gu_layout)
            if cfg_d:
                ops.gemm_parts(a, self._w(lw, "wd"), part, cfg_d[2], cfg_d[1], cfg_d[0], cfg_d[3], cfg_d[4], cfg_d[5])
                r_parts = cfg_d[2]
            else:
                torch.matmul(a, self._row(lw, "wd").t(), out=r)
                r_parts = 0
        if n_sel == 0:                                             # cache-filling chunk of a long prefill
            return None
        # row-pruned tail in one launch: final residual add (the last MLP's split-K partials or r) + RMSNorm of the selected rows only
        if r_parts:
            hn = ops.add_rmsnorm_rows(x, sel_rows, n_sel, self.norm_w, self.eps, part=part, n_parts=r_parts)
        else:
            hn = ops.add_rmsnorm_rows(x, sel_rows, n_sel, self.norm_w, self.eps, r=r)
        cfg_lm = self._tune("lm_head", n_sel) if (self.custom_gemm and n_sel <= self.ROW_CLASSES[-1] and self.V % 8 == 0) else None
        w_lm = self._lm_kt if self._lm_kt is not None else self._lm_head
        if cfg_lm and argmax_out is not None and n_sel <= 128 and cabi.debug("fuse_tail", "1") != "0":
            nb = (self.V + cfg_lm[1] - 1) // cfg_lm[1]
            pairs = torch.empty(n_sel * nb * 2, dtype=torch.float32, device=self.device)
            ops.gemm_argmax(hn, w_lm, argmax_out, pairs, bn=cfg_lm[1], mb=cfg_lm[0], mt=cfg_lm[3], nt=cfg_lm[4], ring=cfg_lm[5])
            return None
        if cfg_lm:
            logits = torch.empty(n_sel, self.V, dtype=self.dtype, device=self.device)
            ops.gemm_skinny(hn, w_lm, out=logits, n_split=1, bn=cfg_lm[1], mb=cfg_lm[0], mt=cfg_lm[3], nt=cfg_lm[4], ring=cfg_lm[5])
        else:
            logits = torch.matmul(hn, self._lm_head.t())
        if argmax_out is not None:
            ops.argmax_rows(logits, out=argmax_out)
            return None
        return logits

    # ---- prefill: plain causal rows over the growing cache ---------------------------------------------
    @torch.no_grad()
    @_on_device
    def prefill(self, ids: Sequence[int], rows: Sequence[int], P0: int = 0, pos: Optional[Sequence[int]] = None):
        """Feeds `ids` (prompt + first window level; lade/models/modeling_llama.py:124-130 plain causal mask, :1527
        `is_prefill`) on top of P0 cached rows as causal chunks of <= max_T tokens.  Only the last chunk runs lm_head,
        and only on `rows` (indices into `ids`; the reference runs it over the whole prompt, :1541-1544), so the last
        chunk is sized to hold every requested row.  pos: positions of the tokens (default P0, P0+1, ...).
        Returns (logits [len(rows), V] in the model dtype, done) with done = tokens fed by the cache-only chunks
        (the last chunk is the step `T = len(ids) - done` at `P = P0 + done`).  The one prefill loop of the package:
        the greedy / sampling / lookahead-parallel loops and hf.jforward_multilevel all come through here."""
        total = len(ids)
        if total == 0 or not rows:
            raise cabi.LadeHipError("prefill needs at least one token and one logits row")
        need = total - min(rows)                                  # the last chunk must contain every requested row
        last_len = min(total, max(self.max_T, need))
        if last_len > self.max_T:
            raise cabi.LadeHipError(f"prefill needs logits of the last {need} tokens in one chunk > engine.max_T={self.max_T}")
        if P0 + total > self.S_max:
            raise cabi.LadeHipError(f"prefill of {total} tokens at P={P0} exceeds the KV cache (S_max={self.S_max})")
        dev = self.device
        t_ids = torch.tensor([int(t) for t in ids], dtype=torch.int32).to(dev, non_blocking=True)
        t_pos = (torch.arange(P0, P0 + total, dtype=torch.int32) if pos is None else torch.tensor([int(t) for t in pos], dtype=torch.int32)).to(dev, non_blocking=True)
        none_sel = torch.zeros(1, dtype=torch.int32, device=dev)
        done = 0
        while total - done > last_len:
            n = min(self.max_T, total - last_len - done)
            self.forward(t_ids[done:done + n], t_pos[done:done + n], StepMask(T=n, P=P0 + done, is_prefill=True), none_sel, 0, rope_len=P0 + total)
            done += n
        sel = torch.tensor([int(r) - done for r in rows], dtype=torch.int32).to(dev, non_blocking=True)
        T = total - done
        # (rope_len: under dynamic NTK scaling the reference sees the whole prompt in ONE forward, kv_seq_len = P0 + total; the chunks must too)
        logits = self.forward(t_ids[done:], t_pos[done:], StepMask(T=T, P=P0 + done, is_prefill=True), sel, len(rows), rope_len=P0 + total)
        return logits, done

    # ---- plain causal decoding on the same kernels (the sequence lookahead must reproduce) ------
    @torch.no_grad()
    @_on_device
    def plain_greedy(self, prompt: Sequence[int], max_length: int, eos_token_id: Optional[int] = None) -> List[int]:
        self.reset()
        ids = [int(t) for t in prompt]
        if len(ids) >= max_length:
            return ids
        logits, _ = self.prefill(ids, [len(ids) - 1])                # long prompts: causal chunks of <= max_T tokens
        P = len(ids)
        one_id = torch.zeros(1, dtype=torch.int32, device=self.device)
        one_pos = torch.zeros(1, dtype=torch.int32, device=self.device)
        sel = torch.zeros(1, dtype=torch.int32, device=self.device)
        while True:
            nxt = int(ops.argmax_rows(logits)[0].item())
            ids.append(nxt)
            if len(ids) >= max_length or (eos_token_id is not None and nxt == eos_token_id):
                break
            one_id.fill_(nxt)
            one_pos.fill_(P)
            logits = self.forward(one_id, one_pos, StepMask(T=1, P=P, is_prefill=True), sel, 1)
            P += 1
        return ids
