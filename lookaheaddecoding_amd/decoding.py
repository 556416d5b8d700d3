"""Lookahead decode loops on the MI355X step engine.

Mirrors the reference's `lade/decoding.py`: `greedy_search_proxy` / `sample_proxy` (:15-34),
`jacobi_greedy_search_multilevel` (:697-1259) and `jacobi_sample_multilevel` (:137-692), with the
same module globals `CONFIG_MAP` / `FUNC_MAP` (:11-12).  What differs is where the work happens:
the window, the n-gram pool, verification, the window roll and the KV commit live on the GPU
(`liblade_hip.so`); the host loop only launches a step, reads back one small record
(`max_hit`, accepted tokens, next candidate count) and applies the stopping criteria.
"""
from __future__ import annotations

import os
import random
import sys
import time
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch

from . import cabi, ops
from .cabi import (CTL_FILL_LEVEL, CTL_G, CTL_HITS, CTL_LST_POS, CTL_LST_TOKEN, CTL_N_INPUT, CTL_P, CTL_WLEN, CTL_WORDS,
                   REC_WORDS, call, ptr, record_seal)
from .engine import StepEngine, _on_device
from .ops import StepMask

FUNC_MAP: dict = {}
CONFIG_MAP: dict = {}
COLOR_PRINT = int(os.environ.get("COLOR_PRINT", 0))


@dataclass
class GenOut:
    tokens: List[int]            # prompt + generated, trimmed to max_length like the reference (:1221-1229)
    steps: int
    generated: int
    trace: List[dict] = field(default_factory=list)


class LadeState:
    """Device-resident integer state of one sequence: window [N-1][wcap], pool [V][G][gs], control block."""

    def __init__(self, V: int, W: int, N: int, G: int, device, max_T: int):
        if not (3 <= N <= cabi.MAX_LEVEL):
            raise cabi.LadeHipError(f"LEVEL={N} unsupported (3..{cabi.MAX_LEVEL}; the reference itself needs LEVEL >= 3)")
        if not (0 <= G <= cabi.MAX_GUESS_SET):
            raise cabi.LadeHipError(f"GUESS_SET_SIZE={G} unsupported (0..{cabi.MAX_GUESS_SET}; <= 0 means no verification branch)")
        if W < 1 or W + N - 3 > cabi.MAX_WINDOW:
            raise cabi.LadeHipError(f"WINDOW_SIZE={W} unsupported")
        self.V, self.W, self.N, self.G, self.gs = V, W, N, G, N - 1
        self.wcap = W + N - 3
        i32 = dict(dtype=torch.int32, device=device)
        self.window = torch.zeros(N - 1, self.wcap, **i32)
        self.ctl = torch.zeros(CTL_WORDS, **i32)
        self.pool_tok = torch.zeros(V, max(G, 1), self.gs, **i32)
        self.pool_cnt = torch.zeros(V, **i32)
        self.guess = torch.zeros(max(G, 1) * self.gs, **i32)
        self.tail = torch.zeros(N + 2, **i32)
        self.ids = torch.zeros(max_T, **i32)
        self.pos = torch.zeros(max_T, **i32)
        self.sel = torch.zeros(1 + self.wcap + G * self.gs, **i32)
        self.am = torch.zeros(1 + self.wcap + G * self.gs, **i32)
        self.record = torch.zeros(REC_WORDS, **i32)
        self.record_host = torch.zeros(REC_WORDS, dtype=torch.int32).pin_memory()
        self.record_host_np = self.record_host.numpy()          # the same pinned bytes, for the polling read (poll_record)
        self.device = device

    def reset(self, window0: Sequence[int], n_prompt: int, prompt_tail: Sequence[int]) -> None:
        N = self.N
        ctl = [0] * CTL_WORDS
        ctl[CTL_N_INPUT] = n_prompt
        ctl[CTL_LST_POS] = n_prompt - 1
        ctl[CTL_WLEN] = len(window0)
        self.ctl.copy_(torch.tensor(ctl, dtype=torch.int32))
        self.window.zero_()
        self.window[0, :len(window0)] = torch.tensor(list(window0), dtype=torch.int32)
        self.pool_cnt.zero_()
        t = list(prompt_tail)[-N:]
        self.tail.copy_(torch.tensor([len(t)] + t + [0] * (N + 1 - len(t)), dtype=torch.int32))

    def read_record(self) -> List[int]:
        self.record_host.copy_(self.record, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return self.record_host.tolist()

    POLL_TIMEOUT_S = float(os.environ.get("LADE_POLL_TIMEOUT_MS", "250")) * 1e-3
    POLL_READS_PER_YIELD = max(1, int(cabi.debug("poll_reads_per_yield", "256")))

    def poll_record(self, step_no: int, timeout_s: Optional[float] = None) -> Optional[List[int]]:
        """The record of step `step_no` as `lade_greedy_post_step` stored it into the pinned host buffer (mapped into the device: no
        copy node, no stream synchronisation).  The host spins on the buffer until the record carries that step number and its seal
        (`lade_record_seal` over the other words) matches what was read - a stale or half-landed record fails one of the two; the seal is
        the ONLY ordering guarantee (the kernel issues no system-scope fence behind the stores).
        The spin is bounded by WALL-CLOCK time (LADE_POLL_TIMEOUT_MS, default 250 ms - far beyond any step) and yields the GIL every 256
        reads, so that a streamer thread or lookahead-parallel rank threads of this process are not starved while this one waits.  None
        when the record does not arrive in time - e.g. pinned memory the device's stores do not reach coherently (HIP_HOST_COHERENT=0);
        the caller then synchronises the stream, reads the record the slow way and STOPS polling for good (LookaheadDecoder._step_graph)."""
        buf = self.record_host_np
        deadline = time.monotonic() + (self.POLL_TIMEOUT_S if timeout_s is None else timeout_s)
        while True:
            for _ in range(self.POLL_READS_PER_YIELD):
                if buf[7] == step_no:
                    rec = buf.tolist()
                    if rec[7] == step_no and (rec[REC_WORDS - 1] & 0xFFFFFFFF) == record_seal(rec, step_no):
                        return rec
            if time.monotonic() > deadline:
                return None
            time.sleep(0)                      # release the GIL: other threads of the process get to run


DRAW_WITH_TORCH = cabi.debug("draw_torch") == "1"


class LookaheadDecoder:
    """Greedy / sampling lookahead decoding of one sequence on a `StepEngine`."""

    def __init__(self, engine: StepEngine, W: int, N: int, G: int, pool_from_prompt: bool = False, lp=None, use_graph: bool = False):
        self.e = engine
        self.use_graph = bool(use_graph)
        self._graph = None
        # GUESS_SET_SIZE <= 0 (-1 = "unlimited" in the reference's README): both reference loops gate the verification
        # branch on `GUESS_SET_SIZE > 0` (lade/decoding.py:402, :948), so the pool is filled but never read and every
        # step accepts exactly one token.  Reproduced as G = 0: no pool, no candidates.
        G = max(int(G), 0)
        self.W, self.N, self.G, self.gs = W, N, G, N - 1
        self.pool_from_prompt = bool(pool_from_prompt)
        self.lp = lp                      # lookahead-parallel context (parallel.LPContext) or None
        self.poll = os.environ.get("LADE_POLL", "1") != "0"       # steady hipGraph steps: poll the host-mapped record instead of synchronising
        self._step_no = 0                 # device step counter (ctl[STEP]) as of the last record read
        self.device = engine.device
        self.st = LadeState(engine.V, W, N, G, engine.device, engine.max_T)
        if self.max_step_tokens() > engine.max_T:
            raise cabi.LadeHipError(f"W={W} N={N} G={G} needs {self.max_step_tokens()} tokens per step > engine.max_T={engine.max_T}")
        # every kernel decision the steady steps of this configuration need (GEMM shapes, the in-step pass, the attention launch
        # parameters) is taken NOW, not inside the first live step of each row class (a multi-second stall for that request)
        if lp is None and os.environ.get("LADE_PREPARE", "1") != "0":
            engine.prepare(sorted({(N - 1) * (W + b) for b in self._buckets()}))

    def max_step_tokens(self) -> int:
        return self.gs + (self.N - 1) * self.W + self.G * self.gs

    # ---- helpers --------------------------------------------------------------------------------
    def _level_sizes(self, fill_level: int) -> List[int]:
        """Lengths of window levels 0..fill_level before the step with that fill level
        (init W+N-3 at lade/decoding.py:902; each fill step trims one: :1040, :1050-1051)."""
        W, N = self.W, self.N
        if fill_level == 0:
            return [W + N - 3]
        if fill_level >= N - 2:
            return [W - 1] + [W] * (N - 2)
        return [W + N - 3 - fill_level] + [W + N - 2 - fill_level] * fill_level

    def _set_sel(self, rows: List[int]) -> int:
        """Device copy of the logits-row selection.  The selections of a run are a handful of shapes (one per phase / candidate
        count), so each is uploaded once and kept: a steady eager step issues no host-to-device copy for it."""
        key = (rows[0], rows[1] if len(rows) > 1 else -1, len(rows), rows[-1])
        cache = self.__dict__.setdefault("_sel_cache", {})
        t = cache.get(key)
        if t is None:
            if len(cache) > 256:
                cache.clear()
            t = cache[key] = torch.tensor(rows, dtype=torch.int32, device=self.e.device)
        self.st.sel[:len(rows)].copy_(t, non_blocking=True)
        return len(rows)

    # ---- stepwise API (bench.py drives single steps; greedy() is start + step until done) ----------
    @torch.no_grad()
    @_on_device
    def start(self, prompt: Sequence[int], eos_token_id: Optional[int] = None, rng: Optional[random.Random] = None) -> None:
        e, st = self.e, self.st
        W, N, G, gs = self.W, self.N, self.G, self.gs
        rng = rng if rng is not None else random
        self.prompt = [int(t) for t in prompt]
        # window init: W+N-3 random prompt tokens (lade/decoding.py:887-902, `copy_from`)
        self.window0 = [rng.choice(self.prompt) for _ in range(W + N - 3)]
        e.reset()
        st.reset(self.window0, len(self.prompt), self.prompt)
        if self.pool_from_prompt:                                   # :915-916
            pt = torch.tensor(self.prompt, dtype=torch.int32, device=e.device)
            call("lade_pool_fill_prompt", ptr(st.pool_tok), ptr(st.pool_cnt), st.V, G, gs, ptr(pt), len(self.prompt))
        self.eos = -1 if eos_token_id is None else int(eos_token_id)
        self.tokens = list(self.prompt)
        self.steps, self.P, self.g, self.fill_level = 0, 0, 0, 0
        self._step_no = 0                  # st.reset zeroed ctl[STEP]
        self.finished_by_eos = False

    # ---- steady step as ONE hipGraph ------------------------------------------------------------------
    # Fixed shapes: a graph is captured per candidate-count bucket gcap; a step with g candidates replays the
    # smallest bucket >= g and feeds T = (N-1)(W+gcap) tokens - the gcap-g unused candidate slots are padded
    # with token 0 (they only see themselves and the input token, and verify ignores them).  The cache length
    # P is read by the kernels from the control block (dyn_P), so the same graphs serve every step: a step is
    # one graph launch plus the read-back of the 24-word record.
    # The buckets follow the GEMM row classes of the engine (32-row activation blocks: a step of 61..64 rows
    # costs what one of 60 does, one of 65 pays for 96): bucket = the most candidates that still fit each
    # class (+ G/4).  Config 2 (W=15 N=5 G=15): {0, 1, 4, 9, 15} = 60 / 64 / 76 / 96 / 120 rows - a step that carries ONE
    # candidate (the common case while n-grams are being accepted) runs at the cold step's cost instead of a
    # 76-row step's (+12 %).  Beyond the last class (T > 256: library GEMMs) the old quartiles {0, G/4, G/2, G}.
    def _buckets(self) -> List[int]:
        G, W, n1 = self.G, self.W, self.N - 1
        classes = [c for c in getattr(self.e, "ROW_CLASSES", ()) if c >= n1 * W]
        if not classes or G == 0:
            return sorted({0, (G + 3) // 4, (G + 1) // 2, G})
        fit = {min(G, c // n1 - W) for c in classes}
        # + G/4: inside one GEMM class fewer padded rows still save attention / lm_head / glue rows (live hot regime of config 2:
        # g = 2..4 at 76 rows 4.33 ms, at 96 rows 4.45)
        return sorted({0, (G + 3) // 4, G} | {b for b in fit if b > 0})

    def _bucket_for(self, g: int) -> int:
        fits = [b for b in self._graphs if b >= g]
        if not fits:        # buckets that no longer fit the cache are not captured (_capture_graphs)
            raise cabi.LadeHipError(f"KV cache exhausted: P={self.P}, a step with {g} candidates no longer fits S_max={self.e.S_max}")
        return min(fits)

    def _graph_body(self, gcap: int, forward_only: bool = False):
        """forward_only (sampling): input assembly + model step + argmax; the fp32 logits of the selected rows stay in
        the graph's memory pool for the host-side verify, the post-step is launched afterwards with the forced hits."""
        e, st = self.e, self.st
        W, N, G, gs = self.W, self.N, self.G, self.gs
        cand_rows = gcap * gs
        mask = StepMask.from_levels(1, self._level_sizes(N - 2), cand_rows, gs, 0)
        T = mask.T
        call("lade_build_inputs", None, None, 1, ptr(st.window), st.wcap, ptr(st.ctl), N - 2, 0, -1, ptr(st.guess), -1, gs, cand_rows,
             ptr(st.ids), ptr(st.pos), None, 0, 1)
        # greedy steps take the row argmax inside the lm_head GEMM (argmax_out: the logits are never materialised); the sampling verify
        # needs the rows themselves
        logits = e.forward(st.ids, st.pos, mask, self._graph_sel[gcap], 1 + W + cand_rows, dyn_P=st.ctl, n_splits=self._graph_splits[gcap],
                           ntk_pad=(st.ctl[CTL_G:CTL_G + 1], gcap, gs),      # (dynamic-NTK RoPE only: the padded candidate slots are not sequence length)
                           argmax_out=None if forward_only else st.am)
        if forward_only:
            ops.argmax_rows(logits, out=st.am)
            return logits.float()                                   # logits.float(), modeling_llama.py:1544
        # LADE_POLL (default on): the post-step stores the sealed record straight into the pinned host buffer and the host polls for it;
        # otherwise a copy node carries it and the host synchronises the stream
        call("lade_greedy_post_step", ptr(st.ctl), ptr(st.window), st.wcap, ptr(st.pool_tok), ptr(st.pool_cnt), st.V, W, N, G,
             ptr(st.am), W, ptr(st.guess), T, cand_rows, 2, int(self.pool_from_prompt), ptr(st.tail), self.eos, None, None, ptr(st.record),
             ptr(st.record_host) if self.poll else None)
        ops.kv_commit(e.kv, 0, 0, 0, ctl=st.ctl)
        if not self.poll:
            st.record_host.copy_(st.record, non_blocking=True)

    def _capture_graphs(self, forward_only: bool = False) -> None:
        e, st = self.e, self.st
        W, N, gs = self.W, self.N, self.gs
        self._graphs, self._graph_sel, self._graph_splits, self._graph_T, self._graph_logits = {}, {}, {}, {}, {}
        state = (st.ctl, st.window, st.pool_cnt, st.pool_tok, st.guess, st.tail) + (() if e.ntk_state is None else (e.ntk_state,))
        saved = [t.clone() for t in state]
        for gcap in self._buckets():
            cand_rows = gcap * gs
            T = (N - 1) * W + cand_rows
            if self.P + T > e.S_max:
                # the warm-up below runs the whole step body at the current cache length: a bucket that no longer fits would
                # write its K/V rows past the cache.  Such a bucket can never be replayed either (_step_graph checks), skip it.
                continue
            rows = [0] + list(range(T - cand_rows - W, T - cand_rows)) + list(range(T - cand_rows, T))
            self._graph_sel[gcap] = torch.tensor(rows, dtype=torch.int32, device=e.device)
            self._graph_splits[gcap] = e.n_splits_for(T, min(e.S_max, max(self.P + T, 1024)))
            self._graph_T[gcap] = T
            # warm-up on a side stream (library handles, allocator), then restore the integer state;
            # the K/V rows it wrote lie at >= P and are rewritten by the real step before being read
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._graph_body(gcap, forward_only)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for t, sv in zip(state, saved):
                t.copy_(sv)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._graph_logits[gcap] = self._graph_body(gcap, forward_only)
            self._graphs[gcap] = g
        torch.cuda.synchronize()
        st.record_host.zero_()              # the warm-up passes left records of the step number the first replay will produce: never mistake one for it
        self._graph = "forward" if forward_only else "step"
        self._graph_eos = self.eos
        self._graph_gen = e.generation

    def _step_graph(self) -> dict:
        e, st = self.e, self.st
        if self._graph != "step" or self._graph_eos != self.eos or self._graph_gen != e.generation:
            self._capture_graphs()
        gcap = self._bucket_for(self.g)
        T = self._graph_T[gcap]
        if abs(e.n_splits_for(T, self.P + T) - self._graph_splits[gcap]) >= 2:      # the cache outgrew the captured KV split count
            self._capture_graphs()
        if self.P + T > e.S_max:
            raise cabi.LadeHipError(f"KV cache exhausted: P={self.P} + T={T} > S_max={e.S_max}")
        P_before = self.P
        self._graphs[gcap].replay()
        rec = st.poll_record(self._step_no + 1) if self.poll else None
        if rec is None:
            torch.cuda.current_stream().synchronize()
            if self.poll:
                # The spin ran out.  Two different things end here: a step that merely took longer than the time-out (the host was descheduled,
                # a profiler attached, another process held the GPU) - its record IS in the mapped buffer now that the stream has drained,
                # and polling stays on; or memory the device's stores do not reach coherently - the record is still absent after the
                # synchronise: read the device copy and stop polling for good (every later step would pay the same time-out; the graphs are
                # re-captured with the copy node instead of the mapped store).
                rec = st.poll_record(self._step_no + 1, timeout_s=0.0)
                if rec is None:
                    rec = st.read_record()
                    self.poll = False
                    self._graph = None
                    print("[lade] the step record did not arrive through the host-mapped buffer (absent even after the stream had drained): "
                          "falling back to stream synchronisation for this decoder", file=sys.stderr, flush=True)
            else:
                rec = st.record_host.tolist()
        self._step_no = rec[7]
        self.steps += 1
        max_hit, n_accept, eos_hit, self.g, self.P = rec[0], rec[1], rec[2], rec[3], rec[4]
        accepted = rec[8:8 + n_accept]
        self.tokens += accepted
        self.finished_by_eos = bool(eos_hit)
        return dict(T=T, P_before=P_before, n_input=1, max_hit=max_hit, max_hit_idx=rec[5], accepted=list(accepted),
                    first_guess=rec[6], g_next=self.g, P_after=self.P, phase=2)

    @torch.no_grad()
    @_on_device
    def step(self, keep_trace: bool = False) -> dict:
        """One decode step = one model forward over T tokens + the device-side post-step; returns the
        host-visible record (accepted tokens, max_hit ...)."""
        e, st = self.e, self.st
        W, N, G, gs = self.W, self.N, self.G, self.gs
        prompt, P, g, fill_level = self.prompt, self.P, self.g, self.fill_level
        if self.use_graph and self.steps > 0 and fill_level >= N - 2:
            return self._step_graph()
        if self.steps == 0:                                          # prefill: prompt + L0, plain causal
            phase = 0
            ids_h = prompt + self.window0
            n_inp, cand_rows = len(self.window0), 0
            # rows whose logits are read: the last prompt token and the window (:1578-1606)
            logits, done = e.prefill(ids_h, [len(prompt) - 1] + list(range(len(prompt), len(ids_h))))
            T, P_before = len(ids_h) - done, done
            n_input = len(prompt) - done                             # out row = last prompt token of the last chunk
        else:
            phase = 2 if fill_level >= N - 2 else 1
            n_input = 1
            ls = self._level_sizes(fill_level)
            cand_rows = g * gs if phase == 2 else 0
            mask = StepMask.from_levels(n_input, ls, cand_rows, gs, P)
            T, P_before = mask.T, P
            call("lade_build_inputs", None, None, n_input, ptr(st.window), st.wcap, ptr(st.ctl), min(fill_level, N - 2), 0, -1,
                 ptr(st.guess), g if phase == 2 else 0, gs, cand_rows, ptr(st.ids), ptr(st.pos), None, 0, 1)
            n_inp = ls[-1]
            # rows whose logits are needed: out row, last level's rows, candidate rows (:1578-1606)
            rows = [n_input - 1] + list(range(T - cand_rows - n_inp, T - cand_rows)) + list(range(T - cand_rows, T))
            n_sel = self._set_sel(rows)
            logits = e.forward(st.ids, st.pos, mask, st.sel, n_sel, argmax_out=st.am)
        if logits is not None:                       # the prefill step returns its logits, a decode step has taken the argmax already
            ops.argmax_rows(logits, out=st.am)
        call("lade_greedy_post_step", ptr(st.ctl), ptr(st.window), st.wcap, ptr(st.pool_tok), ptr(st.pool_cnt), st.V, W, N, G,
             ptr(st.am), n_inp, ptr(st.guess), T, cand_rows, phase, int(self.pool_from_prompt), ptr(st.tail), self.eos, None, None, ptr(st.record), None)
        ops.kv_commit(e.kv, 0, 0, 0, ctl=st.ctl)
        rec = st.read_record()
        self._step_no = rec[7]
        self.steps += 1
        max_hit, n_accept, eos_hit, self.g, self.P = rec[0], rec[1], rec[2], rec[3], rec[4]
        accepted = rec[8:8 + n_accept]
        self.tokens += accepted
        if phase != 2:
            self.fill_level += 1
        self.finished_by_eos = bool(eos_hit)
        return dict(T=T, P_before=P_before, n_input=n_input, max_hit=max_hit, max_hit_idx=rec[5], accepted=list(accepted),
                    first_guess=rec[6], g_next=self.g, P_after=self.P, phase=phase)

    @torch.no_grad()
    @_on_device
    def greedy(self, prompt: Sequence[int], max_length: int, eos_token_id: Optional[int] = None,
               rng: Optional[random.Random] = None, keep_trace: bool = False, on_step=None) -> GenOut:
        """`jacobi_greedy_search_multilevel` (lade/decoding.py:697-1259), single GPU or lookahead parallel.
        on_step(accepted_tokens): called after every step with the tokens it accepted (chat printing / HF streamers,
        lade/decoding.py:1179-1200)."""
        if self.lp is not None and (self.lp.R > 1 or self.lp.force):
            from .parallel import greedy_lp
            out = greedy_lp(self, prompt, max_length, eos_token_id, rng, keep_trace, on_step=on_step)
            # the reference logs on rank 0 only (`if DEBUG and LOCAL_RANK == 0`, lade/decoding.py:1231-1235)
            if CONFIG_MAP.get("DEBUG", 0) and self.lp.rank == 0:
                CONFIG_MAP.setdefault("log", []).append([out.generated, out.steps, round(out.generated / max(out.steps, 1), 2)])
            return out
        self.start(prompt, eos_token_id, rng)
        trace: List[dict] = []
        while True:
            info = self.step()
            if keep_trace:
                trace.append(info)
            if on_step is not None:
                on_step(info["accepted"][:max(0, max_length - (len(self.tokens) - len(info["accepted"])))])
            if self.finished_by_eos or len(self.tokens) >= max_length:      # stopping criteria (:1204-1219)
                break
        generated = min(len(self.tokens), max_length) - len(self.prompt)
        out = GenOut(tokens=self.tokens[:max_length], steps=self.steps, generated=generated, trace=trace)
        if CONFIG_MAP.get("DEBUG", 0):
            CONFIG_MAP.setdefault("log", []).append([generated, self.steps, round(generated / self.steps, 2)])
        return out


    # ---- sampling ------------------------------------------------------------------------------------
    def _sampling_buffers(self):
        if getattr(self, "_smp", None) is None:
            e, G, gs = self.e, max(self.G, 1), self.gs
            rows = 1 + G * gs
            dev = e.device
            self._smp = dict(scal=torch.zeros(rows * G, dtype=torch.float32, device=dev), stats=torch.zeros(rows * 2, dtype=torch.float32, device=dev),
                             scal_host=torch.zeros(rows * G, dtype=torch.float32).pin_memory(), guess_host=torch.zeros(G * gs, dtype=torch.int32).pin_memory(),
                             probs=torch.zeros(1, e.V, dtype=torch.float32, device=dev), probs_host=torch.zeros(e.V, dtype=torch.float32).pin_memory(),
                             am_host=torch.zeros(self.W, dtype=torch.int32).pin_memory(),
                             forced_host=torch.zeros(2 + cabi.MAX_LEVEL, dtype=torch.int32).pin_memory())
            self._smp["forced_np"] = self._smp["forced_host"].numpy()
        return self._smp

    def _draw(self, src: torch.Tensor, row: int, temperature: float, struck: Sequence[int], torch_gen: Optional[torch.Generator]):
        """One token from the distribution of logits row `row` (softmax on the device) with the struck drafts removed.  A CUDA
        generator keeps the draw on the device (what the reference does with the model on a GPU): the token is returned as a
        device tensor and reaches the host with the step's record.  Otherwise the one row goes to the host and
        torch.multinomial consumes the CPU generator (reproducible against the CPU-generated reference traces): returns an int."""
        from .sampling import final_distribution, multinomial_one
        b = self._sampling_buffers()
        probs = ops.softmax_rows(src[row:row + 1], temperature, out=b["probs"])[0]
        if torch_gen is not None and torch_gen.device.type == "cuda":
            if DRAW_WITH_TORCH:                   # LADE_DEBUG=draw_torch: torch.multinomial itself, input checks included (the same token, 12 launches more)
                return torch.multinomial(final_distribution(probs, struck), num_samples=1, generator=torch_gen)
            return multinomial_one(final_distribution(probs, struck), torch_gen)
        b["probs_host"].copy_(probs, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return int(torch.multinomial(final_distribution(b["probs_host"].clone(), struck), num_samples=1, generator=torch_gen).item())

    @torch.no_grad()
    @_on_device
    def sample_start(self, prompt: Sequence[int], warp=None, eos_token_id: Optional[int] = None, rng: Optional[random.Random] = None,
                     torch_gen: Optional[torch.Generator] = None) -> None:
        """Begins a sampling run (`sample` = sample_start + sample_step until done; bench.py drives single steps)."""
        e, W = self.e, self.W
        rng = rng if rng is not None else random
        self.start(prompt, eos_token_id, rng)
        if torch_gen is not None and torch_gen.device.type == "cuda" and not getattr(self, "_draw_warm", False):
            # the draw's torch kernels (exponential, divide, argmax, and the strike-and-renormalise ops of a rejected draft) load their code
            # objects on first use: 10-26 ms in the middle of the first step that rejects a draft.  One dry run on a throw-away generator -
            # the caller's generator is not touched
            from .sampling import final_distribution, multinomial_one
            tg = torch.Generator(device=e.device).manual_seed(0)
            tok = multinomial_one(final_distribution(torch.full((e.V,), 1.0 / e.V, dtype=torch.float32, device=e.device), [0, 1]), tg)
            torch.zeros(4, dtype=torch.int32, device=e.device)[1:2].copy_(tok)
            self._draw_warm = True
        self._s = dict(warp=warp, fused_T=1.0 if warp is None else getattr(warp, "fused_temperature", None), rng=rng, torch_gen=torch_gen,
                       old=list(self.prompt), forced=torch.zeros(2 + cabi.MAX_LEVEL, dtype=torch.int32, device=e.device),
                       override=torch.zeros(W, dtype=torch.int32, device=e.device))

    @torch.no_grad()
    @_on_device
    def sample(self, prompt: Sequence[int], max_length: int, warp=None, eos_token_id: Optional[int] = None,
               rng: Optional[random.Random] = None, torch_gen: Optional[torch.Generator] = None, keep_trace: bool = False,
               on_step=None) -> GenOut:
        """`jacobi_sample_multilevel` (lade/decoding.py:137-692), single GPU (the reference has no LP here).

        The model step, input assembly, pool and window stay on the GPU, and so do the probabilities: `lade_softmax_gather`
        hands the host one small table of draft probabilities per step (sampling.py).  What runs on the host is exactly what
        consumes the reference's RNG streams, in the reference's order: `rng.random()` once per verification trial,
        one `torch.multinomial` for the token that ends the step (:484-540), and the `filter_window` draws (:578-580).
        `warp`: a sampling.Warper (temperature / top-k / top-p; temperature alone is applied inside the kernels) or any callable
        mapping fp32 logits rows [r, V] to warped logits (the HF warpers the reference admits, :375-377)."""
        self.sample_start(prompt, warp, eos_token_id, rng, torch_gen)
        trace: List[dict] = []
        while True:
            info = self.sample_step(keep_trace)
            if keep_trace:
                trace.append(info)
            if on_step is not None:
                on_step(info["accepted"][:max(0, max_length - (len(self.tokens) - len(info["accepted"])))])
            if self.finished_by_eos or len(self.tokens) >= max_length:
                break
        generated = min(len(self.tokens), max_length) - len(self.prompt)
        out = GenOut(tokens=self.tokens[:max_length], steps=self.steps, generated=generated, trace=trace)
        if CONFIG_MAP.get("DEBUG", 0):
            CONFIG_MAP.setdefault("log", []).append([generated, self.steps, round(generated / self.steps, 2)])
        return out

    @torch.no_grad()
    @_on_device
    def sample_step(self, keep_trace: bool = False) -> dict:
        """One sampling step: model forward (eager or the forward-only hipGraph), device-side probability table, host-side
        acceptance walk, one draw, device-side post-step."""
        from .sampling import resolve_drafts
        e, st = self.e, self.st
        W, N, G, gs = self.W, self.N, self.G, self.gs
        S = self._s
        warp, fused_T, rng, torch_gen, all_old_tokens, forced, override = S["warp"], S["fused_T"], S["rng"], S["torch_gen"], S["old"], S["forced"], S["override"]
        buf = self._sampling_buffers()
        prompt_l, P, g, fill_level = self.prompt, self.P, self.g, self.fill_level
        P_before = P
        if self.steps == 0:
            phase = 0
            ids_h = prompt_l + self.window0
            n_inp, cand_rows = len(self.window0), 0
            logits, done = e.prefill(ids_h, [len(prompt_l) - 1] + list(range(len(prompt_l), len(ids_h))))
            logits = logits.float()
            ops.argmax_rows(logits, out=st.am)
            T, P_before = len(ids_h) - done, done
        elif self.use_graph and fill_level >= N - 2:
            # steady step: input assembly + model step + argmax replayed as one hipGraph (candidate rows padded to the bucket)
            if self._graph != "forward" or self._graph_gen != e.generation:
                self._capture_graphs(forward_only=True)
            phase, n_inp = 2, W
            gcap = self._bucket_for(g)
            T, cand_rows = self._graph_T[gcap], gcap * gs
            if abs(e.n_splits_for(T, P + T) - self._graph_splits[gcap]) >= 2:
                self._capture_graphs(forward_only=True)
            if P + T > e.S_max:
                raise cabi.LadeHipError(f"KV cache exhausted: P={P} + T={T} > S_max={e.S_max}")
            self._graphs[gcap].replay()
            logits = self._graph_logits[gcap]
        else:
            phase = 2 if fill_level >= N - 2 else 1
            ls = self._level_sizes(fill_level)
            cand_rows = g * gs if phase == 2 else 0
            mask = StepMask.from_levels(1, ls, cand_rows, gs, P)
            T = mask.T
            call("lade_build_inputs", None, None, 1, ptr(st.window), st.wcap, ptr(st.ctl), min(fill_level, N - 2), 0, -1,
                 ptr(st.guess), g if phase == 2 else 0, gs, cand_rows, ptr(st.ids), ptr(st.pos), None, 0, 1)
            n_inp = ls[-1]
            n_sel = self._set_sel([0] + list(range(T - cand_rows - n_inp, T - cand_rows)) + list(range(T - cand_rows, T)))
            logits = e.forward(st.ids, st.pos, mask, st.sel, n_sel).float()    # logits.float(), modeling_llama.py:1544
            ops.argmax_rows(logits, out=st.am)                                    # window levels are filled by argmax (:459, :545)
        step_inputs = {}
        if keep_trace and phase != 0:                            # what the step fed and judged (tests re-derive the probabilities)
            step_inputs = dict(ids=st.ids[:T].tolist(), pos=st.pos[:T].tolist(), level_sizes=list(self._level_sizes(fill_level)),
                               cand_rows=cand_rows, g=g, drafts=st.guess[:g * gs].tolist() if phase == 2 else [])
        # ---- the step's token(s): rejection-sampling verify over the candidates, or a plain draw (:453-540) ----
        verify = phase == 2 and g > 0
        rows = 1 + g * gs if verify else 1
        if fused_T is not None:                              # temperature only: the kernels scale the logits themselves
            src, skip, temp = logits, n_inp, fused_T
        else:
            # top-k / top-p: one device launch over the out row and the candidate rows (lade_warp_rows); HF warper objects and
            # vocabularies beyond that kernel: torch ops on the device
            src = warp.warp_rows(logits, rows, n_inp) if hasattr(warp, "warp_rows") else None
            if src is None:
                picked = logits[0:1] if not verify else torch.cat([logits[0:1], logits[1 + n_inp:1 + n_inp + g * gs]])
                src = warp(picked)
            skip, temp = 0, 1.0
        max_hit_idx = 0
        if verify:
            ops.softmax_gather(src, rows, skip, st.guess, g, gs, max(G, 1), temp, buf["scal"], buf["stats"])
            buf["scal_host"].copy_(buf["scal"], non_blocking=True)
            buf["guess_host"].copy_(st.guess, non_blocking=True)
            torch.cuda.current_stream().synchronize()                      # the step's one table read-back
            table = buf["scal_host"].view(-1, max(G, 1))[:rows].tolist()
            if keep_trace:
                step_inputs["table"] = table
            verdict = resolve_drafts(table, buf["guess_host"][:g * gs].tolist(), g, gs, rng.random)
            hits, max_hit_idx = list(verdict.accepted), verdict.winner
            if verdict.final_row is not None:
                hits.append(self._draw(src, 0 if verdict.final_row == 0 else verdict.final_row + skip, temp, verdict.struck, torch_gen))
        else:
            hits = [self._draw(src, 0, temp, (), torch_gen)]
        max_hit = len(hits) - 1
        drawn_on_device = hits[-1] if torch.is_tensor(hits[-1]) else None
        if drawn_on_device is not None:
            hits[-1] = 0
        level_override = None
        if phase == 2 and self.eos >= 0:                                      # filter_window on the new level (:578-580)
            buf["am_host"].copy_(st.am[1:1 + W], non_blocking=True)
            torch.cuda.current_stream().synchronize()
            repl = [rng.choice(all_old_tokens) if tok == self.eos else -1 for tok in buf["am_host"].tolist()]
            if any(x >= 0 for x in repl):
                override.copy_(torch.tensor(repl, dtype=torch.int32))
                level_override = override
        # pinned staging (a pageable source would make the copy wait for the draw's kernels on the host); the previous step's copy is long
        # done: every step ends with the record's stream synchronisation
        buf["forced_np"][:] = [max_hit, max_hit_idx] + hits + [0] * (cabi.MAX_LEVEL - len(hits))
        forced.copy_(buf["forced_host"], non_blocking=True)
        if drawn_on_device is not None:
            forced[2 + max_hit:3 + max_hit].copy_(drawn_on_device)          # the drawn token never left the device
        call("lade_greedy_post_step", ptr(st.ctl), ptr(st.window), st.wcap, ptr(st.pool_tok), ptr(st.pool_cnt), st.V, W, N, G,
             ptr(st.am), n_inp, ptr(st.guess), T, cand_rows, phase, int(self.pool_from_prompt), ptr(st.tail), self.eos,
             ptr(forced), ptr(level_override), ptr(st.record), None)
        ops.kv_commit(e.kv, 0, 0, 0, ctl=st.ctl)
        rec = st.read_record()
        self._step_no = rec[7]
        self.steps += 1
        n_accept, eos_hit, self.g, self.P = rec[1], rec[2], rec[3], rec[4]
        accepted = rec[8:8 + n_accept]                                     # = hits[:n_accept]; the record also carries a device-drawn token
        self.tokens += accepted
        all_old_tokens += accepted
        if phase != 2:
            self.fill_level += 1
        self.finished_by_eos = bool(eos_hit)
        return dict(T=T, P_before=P_before, max_hit=max_hit, max_hit_idx=max_hit_idx, accepted=list(accepted), phase=phase, g_next=self.g,
                    P_after=self.P, **step_inputs)
