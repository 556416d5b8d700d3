"""Which kernel runs a projection / the attention launch at a given row class: the isolated autotune pass, the in-step pass that re-takes its
decisions inside a real forward, the persisted decision tables (LADE_TUNE_FILE, the table shipped per GPU model) and the table lookahead-parallel
ranks exchange.  A mixin of `StepEngine` (engine.py): everything here reads and writes the engine's `gemm_cfg` / `attn_cfg` / `_refined`; `forward()` only
looks the decisions up.

Reference: none of this exists there - the reference runs whatever `nn.Linear` / `torch.matmul` dispatches to (lade/models/modeling_llama.py:360-380,
492-494, 558); the decisions pick among the hand-written weight-streaming GEMM's shapes (csrc/gemm_kernel.hpp) and the library."""
from __future__ import annotations

import json
import os
import sys
from typing import Optional, Sequence

import torch

from . import cabi, ops
from .ops import StepMask


def gpu_identity(device) -> tuple:
    """(ISA name, CU count) - what a kernel decision is valid for.  Not the marketing name: torch reports the same MI355X as "AMD Instinct MI355X" or
    "AMD Radeon Graphics" depending on the box's device-id table."""
    p = torch.cuda.get_device_properties(device)
    return str(getattr(p, "gcnArchName", "") or torch.cuda.get_device_name(device)).split(":")[0], int(p.multi_processor_count)


def shipped_tune_table(device) -> str:
    """path of the decision table shipped for this GPU model (it may not exist)"""
    arch, n_cu = gpu_identity(device)
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", f"{arch}_{n_cu}cu.json")


_TUNE_CACHE: dict = {}
_TUNE_TIMES: dict = {}
_TUNE_RANKED: dict = {}
_TUNE_LOCK = __import__("threading").Lock()
_STEP_TUNE_CACHE: dict = {}
_STEP_TUNE_LOCK = __import__("threading").Lock()
REP_ROWS = {32: 30, 64: 60, 96: 92, 128: 128, 160: 156, 192: 180, 256: 240}        # the rows a row class is timed at


class KernelDecisions:
    """mixin of StepEngine: kernel decisions per (projection | attention launch, row class)"""

    def attn_choice(self, T: int):
        """(fused RoPE, work-group rows, split mode) of the attention launch of a T-row step.  Decided once per row class - inside a step
        where the engine can probe one (_refine_in_step) - so that eager steps, hipGraph steps and lookahead-parallel replicas all launch
        the same kernels (16-bit partials round alike)."""
        if self.dtype == torch.float32:
            return (0, 0, 0)
        mclass = next((c for c in self.ROW_CLASSES if T <= c), None)
        if mclass is None or not self.custom_gemm:
            return (0, 128, 0)
        if mclass not in self._refined and not self._refining and not torch.cuda.is_current_stream_capturing():
            for n in self.LAYER_GEMMS:
                self._tune(n, mclass)
            self._refine_in_step(mclass)
        return self.attn_cfg.get(mclass, self.attn_default)

    def n_splits_for(self, T: int, S_tot: int, choice=None) -> int:
        if self.dtype == torch.float32:
            return 1
        forced = int(cabi.debug("attn_splits", "0"))        # experiments only
        if forced > 0:
            return min(forced, self.max_splits, max(1, (S_tot + 63) // 64))
        # always the split + merge form, and never fewer splits than a 1024-key cache would get: the hipGraph steps are
        # captured with exactly that rule, and an eager step of the same (short) sequence must round the same way
        # (16-bit partials) for the two modes to produce the same token stream
        _fuse, wg, mode = choice if choice is not None else self.attn_choice(T)
        return min(ops.choose_splits(self.H, self.H // self.Hkv, T, max(S_tot, 1024), self.n_cu, allow_single=False,
                                     block_rows=0 if wg in (0, 128) else wg, mode=mode), self.max_splits)

    # ---- persisted kernel decisions (LADE_TUNE_FILE) -----------------------------------------------------------
    TUNE_FILE_VERSION = 1

    def _tune_key(self) -> str:
        """what a persisted decision table is valid for: the model shape, the dtype, the weight layouts the kernels stream, the GPU and the probe
        context - NOT this engine's buffer sizes (S_max, max_T): two differently sized engines of one model share a table (the in-step probe
        runs at a fixed context; an engine whose cache is smaller than it has its own key, as it has its own in-process decision)"""
        return json.dumps([self.hidden, self.inter, self.H, self.Hkv, self.d, self.L, self.V, str(self.dtype), list(self.kt_names), self.gu_layout,
                           bool(self.ktile_only), min(self.STEP_TUNE_CONTEXT, self.S_max), self._step_tunable()])

    def _tune_header(self) -> dict:
        return {"version": self.TUNE_FILE_VERSION, "abi": cabi.ABI_VERSION, "device": gpu_identity(self.device)[0], "n_cu": self.n_cu,
                "row_classes": list(self.ROW_CLASSES)}

    def _load_tune_file(self, path: str, strict: bool = True) -> None:
        """LADE_TUNE_FILE=<json>: the kernel decisions of every row class the file holds for this model are ADOPTED instead of measured - the
        multi-second prepare() is skipped and, above all, every process (and every box) that reads the same file launches the same kernels, so
        their 16-bit token streams are the same (a table tuned per box picks other split counts, which round differently).  A file written for
        another GPU model, library ABI or row-class set is refused; a file without an entry for this model is extended when this engine tunes."""
        if not os.path.exists(path):
            return
        try:
            with open(path) as f:
                doc = json.load(f)
        except Exception as e:
            if not strict:
                return
            raise cabi.LadeHipError(f"LADE_TUNE_FILE={path}: unreadable ({e})")
        hdr = self._tune_header()
        if doc.get("header") != hdr:
            if not strict:              # the shipped table of another library generation / GPU: ignored, this process tunes
                return
            raise cabi.LadeHipError(f"LADE_TUNE_FILE={path} was written for {doc.get('header')}, this process is {hdr}: refusing to adopt it "
                                    f"(delete the file or point LADE_TUNE_FILE elsewhere to tune afresh)")
        ent = doc.get("models", {}).get(self._tune_key())
        if not ent:
            return
        for m_s, row in ent.items():
            m = int(m_s)
            if m not in self.ROW_CLASSES:
                continue
            for n in self.GEMM_NAMES:
                if n in row:
                    v = row[n]
                    self.gemm_cfg[(n, m)] = None if v is None else tuple(int(x) for x in v)
            if "attn" in row:
                self.attn_cfg[m] = tuple(int(x) for x in row["attn"][:3])
            if all(n in row for n in self.LAYER_GEMMS):
                self._refined.add(m)
                self.tune_loaded.append(m)
        if self.tune_loaded:
            self.tune_source = path

    def save_tune_file(self, path: Optional[str] = None) -> Optional[str]:
        """Writes (merges) this engine's decisions into the tune file: every row class whose four layer projections are decided (isolated pass +
        in-step pass), + the lm_head decisions taken so far.  Atomic (temp file + rename); other models' entries are kept."""
        path = path or self.tune_file
        if not path or not self.custom_gemm:
            return None
        hdr = self._tune_header()
        doc = {"header": hdr, "models": {}}
        if os.path.exists(path):
            try:
                with open(path) as f:
                    old = json.load(f)
                if old.get("header") == hdr:
                    doc = old
            except Exception:
                pass
        ent = doc.setdefault("models", {}).setdefault(self._tune_key(), {})
        for m in self.ROW_CLASSES:
            row = ent.get(str(m), {})
            if m in self._refined and all((n, m) in self.gemm_cfg for n in self.LAYER_GEMMS):
                for n in self.LAYER_GEMMS:
                    c = self.gemm_cfg[(n, m)]
                    row[n] = None if c is None else list(c)
                row["attn"] = list(self.attn_cfg.get(m, self.attn_default))
            if ("lm_head", m) in self.gemm_cfg:
                c = self.gemm_cfg[("lm_head", m)]
                row["lm_head"] = None if c is None else list(c)
            if row:
                ent[str(m)] = row
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
        os.replace(tmp, path)
        return path

    # ---- GEMM selection ---------------------------------------------------------------------------
    def _tune(self, name: str, M: int):
        """(mb, bn, n_split, mt, nt, ring) of the split-K GEMM for projection `name` at a step of M rows, or None when the library
        GEMM is faster.  Timed once per (projection, row class) on this GPU, rotating through the layers' weights so the
        stream comes from HBM rather than from the Infinity Cache."""
        mclass = next(c for c in self.ROW_CLASSES if M <= c)
        key = (name, mclass)
        if key in self.gemm_cfg:
            return self.gemm_cfg[key]
        if name == "lm_head":
            ws = ([self._lm_head], [self._lm_kt if self._lm_kt is not None else self._lm_head])
        else:         # (row-major: library GEMM - none when the weights are held K-tile-major only, what the skinny GEMM streams)
            # the library is a candidate for a decode-width step only when EVERY layer still holds the row-major weight (a layer whose
            # original was released would pay a rebuild per call)
            rows = [lw[name] for lw in self.layers if name in lw]
            ws = (rows if len(rows) == self.L else None, [self._w(lw, name) for lw in self.layers])
        N, K = self._nk(ws[1][0])
        # one decision per (shape, row class, dtype) and process: engines of the same model (lookahead-parallel ranks run as
        # threads, a decoder rebuilt on the same weights) must pick the same kernel, or their 16-bit results round differently
        # (the gate/up decision differs from a plain projection of the same shape - SwiGLU tail cost, fused-epilogue variant - and
        # timings taken on one GPU model do not transfer to another)
        gkey = (int(N), int(K), mclass, str(self.dtype), name == "wgu", self.gu_layout if name == "wgu" else 0,
                torch.cuda.get_device_name(self.device), self.n_cu, ws[1][0].dim() == 3, ws[0] is None, name == "lm_head")
        with _TUNE_LOCK:
            if gkey not in _TUNE_CACHE:
                self._tuned_ms, self._tuned_ranked = None, []
                _TUNE_CACHE[gkey] = self._tune_timed(name, mclass, ws, N, K)
                _TUNE_TIMES[gkey] = self._tuned_ms
                _TUNE_RANKED[gkey] = self._tuned_ranked
            best = _TUNE_CACHE[gkey]
            self._ranked[key] = _TUNE_RANKED.get(gkey, [])
            # what the winner took when it was timed (isolated launches, every launch on another layer's weights), for bench.py's report
            self.gemm_times[key] = (_TUNE_TIMES.get(gkey), int(N) * int(K) * ws[1][0].element_size())
        self.gemm_cfg[key] = best
        if name == "lm_head" and self.tune_file:
            self.save_tune_file()
        return best

    def _tune_timed(self, name: str, mclass: int, ws, N: int, K: int):
        ws_lib, ws = ws
        a = torch.randn(REP_ROWS[mclass], K, device=self.device).to(self.dtype)
        out = torch.empty(a.shape[0], N, dtype=self.dtype, device=self.device)
        if name == "lm_head":
            return self._tune_lm_head(mclass, a, out, ws_lib, ws, N, K)
        cands = []
        # (m-blocks per work-group, m-blocks per wave, n-tiles per wave (0 = fewest), weight rows per work-group)
        shapes = {32: ((1, 1, 0, (64, 128, 256)), (1, 1, 2, (128, 256)), (2, 1, 0, (128,)), (1, 1, 1, (96,))),
                  64: ((2, 1, 0, (128, 256)), (2, 2, 0, (128, 192, 256)), (2, 2, 2, (192, 256)), (2, 1, 1, (64, 96)), (2, 2, 1, (96,))),
                  96: ((3, 1, 0, (64, 128, 192, 256)), (3, 3, 0, (128, 192, 256)), (3, 3, 2, (192, 256)), (4, 1, 0, (128, 192)), (4, 2, 0, (128, 192)),
                       (3, 3, 1, (96,))),
                  128: ((4, 1, 0, (64, 128, 192, 256)), (4, 2, 0, (128, 192, 256)), (4, 4, 0, (192, 256)), (4, 4, 2, (192, 256)), (4, 4, 1, (96,)),
                        (4, 2, 1, (96,))),
                  # 160 rows (round 6; config 4's steps with 2..6 candidates, 132..156 rows, padded to 192 before): one wave holds all five
                  # m-blocks (5 is prime), the waves lie along N - up to 256 weight rows beside the 20 KB activation tile, three stages
                  160: ((5, 5, 1, (96, 128, 192, 256)), (5, 5, 2, (128, 256))),
                  # 192 / 256 rows (config 4's 120 + 6g-token steps, hot-regime steps): the activation tile alone is 24 / 32 KB per stage,
                  # so the weight tile stays at <= 128 rows for the 3-stage ring to fit the 160 KB of LDS
                  192: ((6, 3, 1, (64, 128)), (6, 3, 2, (128,)), (6, 2, 1, (64,)), (6, 2, 2, (128,))),
                  256: ((8, 4, 1, (64, 128)), (8, 4, 2, (128,)), (8, 2, 1, (64,)), (8, 2, 2, (128,)))}[mclass]
        for mb, mt, nt, bns in shapes:
            for bn in bns:
                nblk = (N + bn - 1) // bn
                for S in sorted({max(1, round(self.n_cu / nblk)), max(1, round(self.n_cu * 2 / nblk)), max(1, round(self.n_cu * 3 / nblk))}):
                    if 2 <= S <= 16 and K // 64 >= 2 * S and S * mclass <= 16 * 128:       # the partial workspace holds 16 x 128 rows (of the CLASS maximum)
                        cands.append((mb, bn, S, mt, nt))

        def time_it(fn, reps=16):
            best_t = float("inf")
            for rnd in range(2):                       # best of two rounds: the winner must not be a timing fluke
                for i in range(2):
                    fn(i)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(reps):
                    fn(i + rnd)
                e1.record()
                torch.cuda.synchronize()
                best_t = min(best_t, e0.elapsed_time(e1) / reps)
            return best_t

        # what follows the GEMM inside the step: a split-K or library gate/up GEMM is followed by the SwiGLU kernel (~6 us with its
        # launch boundary), the fused-epilogue variant (S = 1 below) is not
        # (a dependent launch ~2.5 us + its reads of the GEMM output at ~4 TB/s: 6.3 us measured behind 4 partials of the 7B shape
        # at 60 rows, ~9 us behind 2 partials of the 13B shape at 120 rows; times here are in ms)
        Mrows = a.shape[0]
        tail = (lambda out_bytes: 0.0025 + out_bytes / 4e9) if name == "wgu" else (lambda out_bytes: 0.0)
        if name == "wgu" and cabi.debug("gu_tail_fixed"):          # experiment: the flat 6 us estimate
            tail = lambda out_bytes: 0.006
        if name != "wgu" and cabi.debug("tune_consumer"):          # experiment: the consumer's read of the partials, every projection
            rate = float(cabi.debug("tune_consumer")) * 1e9
            tail = lambda out_bytes: out_bytes / rate
        # + the consumer's extra read; K-tile-only weights: the library would need its row-major operand rebuilt per call - not a candidate
        t_lib = float("inf") if ws_lib is None else time_it(lambda i: torch.matmul(a, ws_lib[i % len(ws_lib)].t(), out=out)) + 0.003 + tail(Mrows * N * 2)
        timed = []                        # (ms incl. the consumer tail, (mb, bn, S, mt, nt, ring))
        for (mb, bn, S, mt, nt) in cands:
            t = time_it(lambda i: ops.gemm_parts(a, ws[i % len(ws)], self.ws_part, S, bn, mb, mt, nt)) + tail(S * Mrows * N * 4)
            timed.append((t, (mb, bn, S, mt, nt, 0)))
        act = None
        if name == "wgu" and self.gu_layout == 1:
            # no split-K: BN weight rows x the whole K per work-group, SwiGLU in the epilogue, output in the model dtype.  Needs
            # N / BN work-groups to cover the CUs on their own: 96-row blocks at the 7B / 13B widths.
            act = torch.empty(a.shape[0], N // 2, dtype=self.dtype, device=self.device)
            mbs = mclass // 32
            for bn in (64, 96, 128) + ((224,) if N % 224 == 0 else ()):        # 224: N = 256 x 224 at the 70B width
                for mt in sorted({1, 2 if mbs % 2 == 0 else 1, mbs if mbs <= 5 else mbs // 2}):
                    if mbs % mt or (mbs // mt) * (bn // 32) > 8:
                        continue
                    try:
                        t = time_it(lambda i: ops.gemm_swiglu(a, ws[i % len(ws)], act, bn, mbs, mt, 1))
                    except cabi.LadeHipError:
                        continue                      # wave grid not built for this row class
                    timed.append((t, (mbs, bn, 1, mt, 1, 0)))
        # second pass: the LDS ring depth of the best few.  More stages = more bytes in flight per work-group, fewer work-groups per CU;
        # which of the two a projection needs depends on its split count (an unsplit gate/up GEMM is one work-group per CU whatever the
        # ring costs, a 512-work-group split-K launch needs two per CU), so the depth is a per-kernel decision like the shape itself
        if cabi.debug("tune_ring", "0") != "0":      # (off: isolated launches do not rank ring depths the way the step does - _refine_in_step decides them)
            for t0, (mb, bn, S, mt, nt, _r) in sorted(timed)[:3]:
                stage_bytes = (bn + 32 * mb) * 128
                for ring in (3, 5, 6, 8):
                    if ring * stage_bytes > 160 * 1024 or ring == min(4, 160 * 1024 // stage_bytes):
                        continue
                    try:
                        if S == 1:
                            t = time_it(lambda i: ops.gemm_swiglu(a, ws[i % len(ws)], act, bn, mb, mt, nt, ring))
                        else:
                            t = time_it(lambda i: ops.gemm_parts(a, ws[i % len(ws)], self.ws_part, S, bn, mb, mt, nt, ring)) + tail(S * Mrows * N * 4)
                    except cabi.LadeHipError:
                        continue
                    timed.append((t, (mb, bn, S, mt, nt, ring)))
        best, t_best = None, t_lib
        if timed and min(timed)[0] < t_lib:
            t_best, best = min(timed)
        self._tuned_ms = t_best
        self._tuned_ranked = sorted(timed) + ([(t_lib, None)] if ws_lib is not None else [])
        if cabi.debug("tune_verbose"):          # tools/gemm_tune_probe.py: what the tuner saw
            mbytes = N * K * ws[0].element_size() / 1e6
            top = " ".join(f"{c}:{t * 1e3:.1f}" for t, c in sorted(timed)[:6])
            print(f"[tune] {name}:{mclass} rows={Mrows} N={N} K={K} lib {t_lib * 1e3:.1f} us | best {best} {t_best * 1e3:.1f} us "
                  f"({mbytes / (t_best * 1e3):.2f} TB/s) | {top}", file=sys.stderr, flush=True)
        return best

    def _tune_lm_head(self, mclass: int, a, out, ws_lib, ws, N: int, K: int):
        """lm_head on the rows that are read (1 for plain decoding, 1 + W + g gs for a lookahead step): the skinny GEMM WITHOUT split-K -
        N / bn work-groups cover the CUs on their own at vocabulary sizes, the output is written once in the model dtype - against
        the library.  Measured at V = 32000, K = 4096, 16 rows: library 57.3 us, skinny on row-major weights 48.3 us, on the K-tile-major
        copy 39.4 us (6.65 TB/s; `profiles/r3_lm_head_probe.txt`)."""
        mbs = mclass // 32

        def time_it(fn, reps=12):
            best_t = float("inf")
            for rnd in range(2):
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best_t = min(best_t, e0.elapsed_time(e1) / reps)
            return best_t

        t_lib = time_it(lambda: torch.matmul(a, ws_lib[0].t(), out=out))
        timed = []
        for bn in (64, 96, 128, 192, 256):
            for mt in sorted({1, mbs if mbs <= 5 else mbs // 2}):
                for nt in (0, 1, 2):
                    if mbs % mt:
                        continue
                    try:
                        t = time_it(lambda: ops.gemm_skinny(a, ws[0], out=out, n_split=1, bn=bn, mb=mbs, mt=mt, nt=nt))
                    except cabi.LadeHipError:
                        continue                      # wave grid not built for this row class
                    timed.append((t, (mbs, bn, 1, mt, nt, 0)))
        if cabi.debug("tune_ring", "0") != "0":
            for t0, (mb, bn, S, mt, nt, _r) in sorted(timed)[:2]:
                stage_bytes = (bn + 32 * mb) * 128
                for ring in (3, 5, 6, 8):
                    if ring * stage_bytes > 160 * 1024 or ring == min(4, 160 * 1024 // stage_bytes):
                        continue
                    try:
                        t = time_it(lambda: ops.gemm_skinny(a, ws[0], out=out, n_split=1, bn=bn, mb=mb, mt=mt, nt=nt, ring=ring))
                    except cabi.LadeHipError:
                        continue
                    timed.append((t, (mb, bn, 1, mt, nt, ring)))
        best, t_best = None, t_lib
        if timed and min(timed)[0] < t_lib:
            t_best, best = min(timed)
        self._tuned_ms = t_best
        if cabi.debug("tune_verbose"):
            top = " ".join(f"{c}:{t * 1e3:.1f}" for t, c in sorted(timed)[:4])
            print(f"[tune] lm_head:{mclass} rows={a.shape[0]} N={N} K={K} lib {t_lib * 1e3:.1f} us | best {best} {t_best * 1e3:.1f} us "
                  f"({N * K * 2 / 1e6 / (t_best * 1e3):.2f} TB/s) | {top}", file=sys.stderr, flush=True)
        return best

    # ---- the same decisions, re-taken inside a step ------------------------------------------------------------------------
    STEP_TUNE_LAYERS = 8
    STEP_TUNE_MIN_BYTES = 512 << 20
    STEP_TUNE_CONTEXT = 2048             # keys of the probe step's cache: a constant, so that the decision does not depend on this engine's S_max

    def _refine_in_step(self, mclass: int, allow_grow: bool = False) -> None:
        """Isolated launches do not rank GEMM configurations the way a step does: back to back on one stream a kernel meets warm caches, no
        consumer reads its split-K partials and no glue kernel sits between it and the next weight stream.  Round 4 measured it: a
        4-stage LDS ring took the 7B step from 3.91 / 3.86 to 3.78 / 3.77 ms (same box, alternating), while the isolated pass ranked 3
        stages first for nearly every projection.  So after the isolated pass has produced a short list per projection, the decision is
        re-taken where it counts: hipGraphs of a real forward over the first STEP_TUNE_LAYERS layers (weights >> the Infinity Cache, the
        attention pair, RoPE and both norms in place, the consumers reading the partials), one graph per candidate - the short list x
        ring depths, plus the best candidate of each smaller split count - replayed round-robin; projection after projection
        (coordinate descent, one pass), a challenger replaces the incumbent only when it wins by > 0.3 %.  Round 5: the ATTENTION launch
        is the fifth coordinate - (RoPE + KV append fused into it or not) x (work-group rows 128 / 64 / 32) x (split rule), the launch
        parameters of lade_attn_args - decided the same way on the same graphs.
        The probe step runs at a FIXED context (STEP_TUNE_CONTEXT keys, or the whole cache of a smaller engine - part of the cache key) on
        cache rows that are saved and restored, and on no live workspace; workspaces too small for the class's probe rows are grown first.
        One decision per model shape, row class, probe context and process; never taken during a stream capture (forward() refuses to
        capture a class whose decision is still open: warm the class up first, or call prepare()).  LADE_TUNE_STEP=0 keeps the isolated table."""
        if mclass in self._refined or self._refining or not self.custom_gemm:
            return
        T = REP_ROWS[mclass]
        esz = self.layers[0]["wo"].element_size() if "wo" in self.layers[0] else self.layers[0]["wo_kt"].element_size()
        layer_bytes = esz * (self.hidden * ((self.H + 2 * self.Hkv) * self.d + self.H * self.d) + 3 * self.hidden * self.inter)
        n_probe = min(self.L, self.STEP_TUNE_LAYERS)
        if not self._step_tunable():
            self._refined.add(mclass)          # toy models sit in the Infinity Cache whatever runs: the isolated table stands (a property of the MODEL, not of this engine's buffers)
            if self.tune_file:
                self.save_tune_file()
            return
        if torch.cuda.is_current_stream_capturing():
            raise cabi.LadeHipError(f"row class {mclass}: the in-step kernel decisions are still open during a stream capture - run the step eagerly once "
                                    f"(or StepEngine.prepare([rows])) before capturing it")
        for n in self.LAYER_GEMMS:
            self._tune(n, mclass)
        if T > self.max_T:
            # the probe wants REP_ROWS rows of workspace.  prepare() (construction time: nobody has captured a graph over the workspaces yet)
            # grows them, so that a decision does not depend on how this engine was sized; a LAZY refine from inside forward() / attn_choice()
            # must not re-allocate buffers under a caller who may hold graphs over them (grow() bumps `generation`, but only
            # LookaheadDecoder looks at it): it probes at the rows the engine has - still rows of this class, since the triggering step is
            if allow_grow:
                self.grow(self.S_max, T)
            else:
                T = self.max_T
        ctx = min(self.STEP_TUNE_CONTEXT, self.S_max)
        skey = (self.hidden, self.inter, self.H, self.Hkv, self.d, self.L, mclass, str(self.dtype), torch.cuda.get_device_name(self.device),
                self.n_cu, tuple(self.kt_names), self.gu_layout, tuple(self.gemm_cfg[(n, mclass)] for n in self.LAYER_GEMMS), ctx, self.attn_default,
                cabi.debug("attn_tune", "0"))
        with _STEP_TUNE_LOCK:
            if skey not in _STEP_TUNE_CACHE:
                _STEP_TUNE_CACHE[skey] = self._refine_timed(mclass, T, n_probe, ctx)
            choice, log = _STEP_TUNE_CACHE[skey]
        for n, c in choice.items():
            if n == "attn":
                self.attn_cfg[mclass] = tuple(c)
                continue
            self.gemm_cfg[(n, mclass)] = c
            ranked = dict((cfg[:5] if cfg else None, t) for t, cfg in reversed(self._ranked.get((n, mclass), [])))
            if (n, mclass) in self.gemm_times:
                self.gemm_times[(n, mclass)] = (ranked.get(c[:5] if c else None, self.gemm_times[(n, mclass)][0]), self.gemm_times[(n, mclass)][1])
        self.step_tune_log[mclass] = log
        self._refined.add(mclass)
        if self.tune_file:
            self.save_tune_file()

    def _step_tunable(self) -> bool:
        """whether this MODEL's decisions are re-taken inside a step: its first STEP_TUNE_LAYERS layers must outweigh the Infinity Cache"""
        if os.environ.get("LADE_TUNE_STEP", "1") == "0" or not self.custom_gemm or not self.layers:
            return False
        lw = self.layers[0]
        esz = (lw["wo"] if "wo" in lw else lw["wo_kt"]).element_size()
        layer_bytes = esz * (self.hidden * ((self.H + 2 * self.Hkv) * self.d + self.H * self.d) + 3 * self.hidden * self.inter)
        return layer_bytes * min(self.L, self.STEP_TUNE_LAYERS) >= self.STEP_TUNE_MIN_BYTES and self.S_max >= 512

    def prepare(self, rows: Sequence[int]) -> None:
        """Takes every kernel decision a step of each of these row counts needs (isolated GEMM pass, the in-step pass, the attention launch
        parameters) NOW - at construction / before the first request - instead of inside the first live forward of each row class, which
        would stall that request for seconds (4 projections + the attention launch x up to 18 graph captures)."""
        if not self.custom_gemm:
            return
        with torch.cuda.device(self.device):
            for T in rows:
                mclass = next((c for c in self.ROW_CLASSES if T <= c), None)
                if mclass is None:
                    continue
                for n in self.LAYER_GEMMS:
                    self._tune(n, mclass)
                self._refine_in_step(mclass, allow_grow=True)

    def _step_candidates(self, name: str, mclass: int):
        ranked = [c for _t, c in self._ranked.get((name, mclass), [])]
        inc = self.gemm_cfg[(name, mclass)]
        shapes = [c for c in ranked[:3] if c is not None]
        for S in sorted({c[2] for c in ranked if c is not None})[:2]:                   # the best candidate of the two smallest split counts
            shapes.append(next(c for c in ranked if c is not None and c[2] == S))
        out = [inc]
        for c in shapes:
            stage_bytes = (c[1] + 32 * c[0]) * 128
            dflt = min(4, 160 * 1024 // stage_bytes)
            # (the depths the kernel is compiled for: 2, 3, 4 = default, 5, 6, 8; a DOUBLE buffer only for the large stages of the 160-row
            # class and wider: two stages beat three on 13B gate/up at 256 weight rows + 160 activation rows, 68.9 vs 71.7 us isolated, and
            # the in-step pass picks them for all four projections of the 13B 192-row class, profiles/r6_rows_curve_13b.txt)
            for ring in (0, 3, 5, 6, 8) + ((2,) if stage_bytes >= 32 * 1024 else ()):
                cand = c[:5] + (ring,)
                if (ring == 0 or (ring != dflt and ring * stage_bytes <= 160 * 1024)) and cand not in out:
                    out.append(cand)
        return out[:18]

    def _attn_candidates(self, mclass: int, T: int, S_tot: int):
        """(fused, work-group rows, split mode) x what they resolve to at the probe context, one entry per distinct launch.
        Round 6: the attention launch is FROZEN at the default (two launches, 128-row work-groups, sqrt split rule) - round 5's in-step
        pass found every candidate within 0.2-1.5 % of it at every BASELINE shape (profiles/r5_attn_tune.txt: inside the bench's own block
        noise), while a box-dependent choice of the split count made the 16-bit token stream and the PMC evidence box-dependent.
        LADE_DEBUG=attn_tune brings the coordinate back (experiments); the fused-RoPE forms are candidates only in a -DLADE_EXPERIMENTAL build."""
        inc = self.attn_cfg.get(mclass, self.attn_default)
        if cabi.debug("attn_tune", "0") != "1":
            return [inc]
        qkv = self.gemm_cfg.get(("wqkv", mclass))
        fused_ok = (cabi.experimental() and qkv is not None and qkv[3] >= 0 and qkv[2] <= 4 and cabi.debug("fuse_rope", "") not in ("", "off"))
        fuses = (1, 0) if fused_ok else (0,)          # (the producer mode, fused = 2, only ever by LADE_DEBUG=fuse_rope=2 as the default: never picked by a tuner)
        rows = (self.H // self.Hkv) * T
        shapes = [128, 64] + ([32] if rows <= 64 or self.H != self.Hkv else [])
        out, seen = [], set()
        for cand in [inc] + [(f, wg, mode) for f in fuses for wg in shapes for mode in (0, 1, 2)]:
            if cand[0] and not fuses[0]:
                continue
            key = (cand[0], cand[1], self.n_splits_for(T, S_tot, choice=cand))
            if key not in seen:
                seen.add(key)
                out.append(tuple(cand))
        return out[:18]

    def _refine_timed(self, mclass: int, T: int, n_probe: int, ctx: int):
        dev = self.device
        P = ctx - T
        gen = torch.Generator(device=dev)
        gen.manual_seed(20240924)
        ids = torch.randint(0, self.V, (T,), device=dev, dtype=torch.int32, generator=gen)
        pos = torch.arange(P, P + T, device=dev, dtype=torch.int32)
        mask = StepMask(T=T, P=P, is_prefill=True)
        sel = torch.zeros(1, dtype=torch.int32, device=dev)
        all_layers = self.layers
        kv_saved = [(self._k_views[li][:, P:P + T].clone(), self._vt_views[li][:, :, P:P + T].clone()) for li in range(n_probe)]
        ntk_saved = None if self.ntk_state is None else self.ntk_state.clone()
        saved_ev, self.attn_events = self.attn_events, None
        saved_attn = self.attn_cfg.get(mclass)
        choice, log = {}, {}
        self._refining = True
        self.layers = all_layers[:n_probe]

        def measure(cands, apply):
            """one graph per candidate (thread-local capture mode: lookahead-parallel ranks may be threads of this process, and another
            rank's allocation or synchronisation must not invalidate this capture), replayed round-robin; best time per layer of each"""
            graphs = []
            for cand in cands:
                apply(cand)
                try:
                    run = lambda: self.forward(ids, pos, mask, sel, 0, n_splits=self.n_splits_for(T, P + T, choice=self.attn_cfg.get(mclass, self.attn_default)))
                    run()                                              # eager once: first-launch attributes, argument validation
                    torch.cuda.synchronize()
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        run()
                except cabi.LadeHipError:
                    continue
                graphs.append((cand, g))
            times = {c: float("inf") for c, _g in graphs}
            for rnd in range(6):
                for cand, g in (graphs if rnd % 2 == 0 else graphs[::-1]):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    g.replay()
                    e1.record()
                    e1.synchronize()
                    if rnd > 0:                                        # round 0 warms every graph up
                        times[cand] = min(times[cand], e0.elapsed_time(e1) / n_probe)
            del graphs
            return times

        try:
            self.attn_cfg[mclass] = self.attn_cfg.get(mclass, self.attn_default)
            for name in self.LAYER_GEMMS:
                inc = self.gemm_cfg[(name, mclass)]

                def apply(cand, name=name):
                    self.gemm_cfg[(name, mclass)] = cand
                times = measure(self._step_candidates(name, mclass), apply)
                best = min(times, key=times.get)
                pick = best if times[best] < times.get(inc, float("inf")) * 0.997 else inc
                self.gemm_cfg[(name, mclass)] = choice[name] = pick
                log[name] = {"isolated_choice": inc, "in_step_choice": pick, "ms_per_layer_isolated_choice": round(times.get(inc, float("nan")), 5),
                             "ms_per_layer_in_step_choice": round(times[pick], 5), "candidates": len(times)}
                if cabi.debug("tune_verbose"):
                    top = " ".join(f"{c}:{t * 1e3:.1f}" for c, t in sorted(times.items(), key=lambda kv: kv[1])[:6])
                    print(f"[tune-step] {name}:{mclass} rows={T} isolated choice {inc} {times.get(inc, float('nan')) * 1e3:.1f} us/layer -> {pick} "
                          f"{times[pick] * 1e3:.1f} | {top}", file=sys.stderr, flush=True)
            # the attention launch: fused RoPE + KV append or not, work-group rows, split rule
            inc = self.attn_cfg[mclass]

            def apply_attn(cand):
                self.attn_cfg[mclass] = cand
            times = measure(self._attn_candidates(mclass, T, P + T), apply_attn)
            if times:
                best = min(times, key=times.get)
                # (a challenger must win by > 1 %: the candidates of this coordinate lie within a fraction of a per cent of each other on most
                # shapes - first round-5 runs flipped between 6 and 8 splits from box to box on 0.2-0.6 % - and more splits mean more partial
                # traffic for nothing)
                pick = best if times[best] < times.get(inc, float("inf")) * 0.99 else inc
                self.attn_cfg[mclass] = choice["attn"] = tuple(pick)
                log["attn"] = {"default": inc, "in_step_choice": pick, "ms_per_layer_default": round(times.get(inc, float("nan")), 5),
                               "ms_per_layer_in_step_choice": round(times[pick], 5), "candidates": len(times), "probe_context": ctx,
                               "n_splits_at_probe": self.n_splits_for(T, P + T, choice=pick),
                               "ranked_us_per_layer": [[list(c), round(t * 1e3, 2)] for c, t in sorted(times.items(), key=lambda kv: kv[1])[:8]],
                               "fields": "(RoPE + KV append fused into the launch, work-group rows, split mode 0 sqrt | 1 fill | 2 half)"}
                if cabi.debug("tune_verbose"):
                    top = " ".join(f"{c}:{t * 1e3:.1f}" for c, t in sorted(times.items(), key=lambda kv: kv[1])[:8])
                    print(f"[tune-step] attn:{mclass} rows={T} default {inc} {times.get(inc, float('nan')) * 1e3:.1f} us/layer -> {pick} {times[pick] * 1e3:.1f} | {top}",
                          file=sys.stderr, flush=True)
        finally:
            self.layers = all_layers
            self._refining = False
            self.attn_events = saved_ev
            if "attn" not in choice:
                if saved_attn is None:
                    self.attn_cfg.pop(mclass, None)
                else:
                    self.attn_cfg[mclass] = saved_attn
            for li, (k, v) in enumerate(kv_saved):
                self._k_views[li][:, P:P + T].copy_(k)
                self._vt_views[li][:, :, P:P + T].copy_(v)
            if ntk_saved is not None:
                self.ntk_state.copy_(ntk_saved)
            torch.cuda.synchronize()
        return choice, log

    LAYER_GEMMS = ("wqkv", "wo", "wgu", "wd")
    GEMM_NAMES = LAYER_GEMMS + ("lm_head",)
    # 32-row activation blocks per work-group; 160 since round 6 (a 129..160-row step padded to 192 before; LADE_DEBUG=row_classes=r5: without it, A/B runs)
    ROW_CLASSES = (32, 64, 96, 128, 192, 256) if cabi.debug("row_classes") == "r5" else (32, 64, 96, 128, 160, 192, 256)

    TUNE_NAMES = GEMM_NAMES + ("attn",)        # rows of the decision table lookahead-parallel ranks exchange (parallel.encode_tune_table)

    def tune_all(self) -> dict:
        """Every (projection, row class) decision of this engine - and the attention launch parameters per row class (`attn:<class>` =
        (fused RoPE, work-group rows, split mode, 0, 0, 0)) - as a plain dict (lookahead parallelism: rank 0 tunes, the other ranks adopt
        its table through `adopt_gemm_cfg`, so that all replicas round alike)."""
        if not self.custom_gemm:
            return {}
        for m in self.ROW_CLASSES:
            for n in self.GEMM_NAMES:
                self._tune(n, m)
            self._refine_in_step(m, allow_grow=True)
        out = {f"{n}:{m}": self.gemm_cfg[(n, m)] for n in self.GEMM_NAMES for m in self.ROW_CLASSES}
        out.update({f"attn:{m}": tuple(self.attn_cfg.get(m, self.attn_default)) + (0, 0, 0) for m in self.ROW_CLASSES})
        return out

    def adopt_gemm_cfg(self, table: dict) -> None:
        for k, v in table.items():
            n, m = k.split(":")
            if n == "attn":
                if v is not None:
                    self.attn_cfg[int(m)] = tuple(int(x) for x in v[:3])
                continue
            self.gemm_cfg[(n, int(m))] = None if v is None else (tuple(v) + (0,))[:6]       # (a 5-tuple of an older table: default ring)
            self._refined.add(int(m))                # an adopted decision is final: every replica must run the kernels rank 0 chose
