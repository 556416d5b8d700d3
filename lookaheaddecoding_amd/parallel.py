"""Lookahead parallelism (LP): the W window columns and the g candidates of a step are sharded over the
R GPUs of a node; every rank holds a full model replica and the KV cache of the accepted prefix.

Reference: lade/decoding.py:905-906 (window broadcast), :956-963 (candidate shard), :973-986 (window
shard), :1023-1024, 1043-1046, 1055-1058, 1088-1107 (per-step object collectives), :1148-1153 (on a
hit the cache is cut back and the accepted tokens are re-fed), lade/lade_distributed.py.

Re-design for RCCL over xGMI: the reference issues four pickled *object* collectives per step (each
a size exchange + payload round on the host).  Here every rank packs ONE fixed-size int32 record on
the device (`lade_lp_pack`: first_guess, its best verification result, its new window columns), the
host issues ONE `all_gather_into_tensor` (RCCL; tens of bytes, latency bound, so a single small
collective on the compute stream), and every rank applies the same deterministic reduction on the
device (`lade_lp_reduce_apply`).  Rank-local state (window, pool, control block) therefore stays
bit-identical on all ranks without further traffic.

The orchestration (`greedy_lp`) is written against a small backend interface so that the partition /
exchange / loop logic can be exercised on CPU with gloo (world_size 2) in the test-suite, where a
test-only backend stands in for the HIP kernels; the product backend is `HipLPBackend` (no CPU path).
"""
from __future__ import annotations

import os
import random
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

REC_HEAD = 4           # first_guess, n_inp, g_local, (reserved)


@dataclass
class LPContext:
    rank: int
    world: int
    group: Optional[object] = None
    force: bool = False        # run the lookahead-parallel code path even with one rank (a 1-GPU box exercising the RCCL path)

    @property
    def R(self) -> int:
        return self.world


def window_shard(window_len: int, R: int, r: int) -> Tuple[int, int]:
    """Columns [c0, c1) of the window owned by rank r; window_len = len(L0) + 1 (lade/decoding.py:974-977)."""
    split = (window_len + R - 1) // R
    return min(split * r, window_len), min(split * (r + 1), window_len)


def guess_shard(g: int, R: int, r: int) -> Tuple[int, int]:
    """Candidates [lo, hi) verified by rank r (lade/decoding.py:958-961)."""
    cnt = (g + R - 1) // R
    return min(cnt * r, g), min(cnt * (r + 1), g)


def shard_level_sizes(level_lens: Sequence[int], c0: int, c1: int) -> List[int]:
    """Level sizes of rank r's step: the whole L0 prefix up to its last column, and its own columns of
    every higher level (lade/decoding.py:981-984)."""
    out = [min(c1 - 1, level_lens[0])]
    for ln in level_lens[1:]:
        out.append(max(min(c1, ln) - min(c0, ln), 0))
    return out


def rec_words(gs: int, wcap: int, G: int = 0) -> int:
    """int32 words of one rank's record: head | new window tokens [wcap] | argmax ids of its candidate rows [<= G*gs]"""
    return REC_HEAD + wcap + max(G, 0) * gs


class FileChannel:
    """A byte channel between the ranks of one node that needs nothing but a shared directory: rank 0 publishes a blob under a
    name (written to a temporary file, then renamed: readers never see a partial blob), the others poll for it.  What a caller
    without torch.distributed hands to `RcclComm` for the 128-byte unique id (the reference's channel is the TCP store behind
    `dist.init_process_group`, lade/utils.py:31)."""

    def __init__(self, directory: str, rank: int, timeout_s: float = 120.0):
        self.dir, self.rank, self.timeout_s = directory, rank, timeout_s
        os.makedirs(directory, exist_ok=True)

    def __call__(self, blob: Optional[bytes], name: str = "rccl_unique_id") -> bytes:
        import time
        path = os.path.join(self.dir, name)
        if self.rank == 0:
            tmp = path + f".tmp{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(blob)
            os.replace(tmp, path)
            return blob
        t0 = time.time()
        while not os.path.exists(path):
            if time.time() - t0 > self.timeout_s:
                raise TimeoutError(f"rank {self.rank}: nothing published at {path} after {self.timeout_s:.0f} s")
            time.sleep(0.01)
        with open(path, "rb") as f:
            return f.read()


class RcclComm:
    """The C ABI's own communicator (lade_lp_comm_create / lade_lp_allgather / lade_lp_comm_count / lade_lp_comm_destroy): what a
    caller without torch.distributed binds.  The 128-byte unique id is produced on rank 0 and handed to the other ranks by
    `exchange_id`, a callable `bytes (rank 0) | None -> bytes` over any host channel the caller has (`FileChannel`, a socket, MPI).
    Only when none is given AND torch.distributed is already initialised is its store used; a world of one rank needs no channel."""

    def __init__(self, rank: int, world: int, exchange_id=None):
        import ctypes as C
        from . import cabi
        self._cabi, self.rank, self.world = cabi, rank, world
        buf = (C.c_char * 128)()
        if rank == 0:
            cabi.call_plain("lade_lp_unique_id", buf)
        ident = bytes(buf)
        if world > 1:
            if exchange_id is None:
                if not (dist.is_available() and dist.is_initialized()):
                    raise cabi.LadeHipError("RcclComm with world > 1 needs `exchange_id` (a byte channel for the 128-byte RCCL unique id, e.g. "
                                            "parallel.FileChannel) when torch.distributed is not initialised")

                def exchange_id(b):
                    box = [b if rank == 0 else None]
                    dist.broadcast_object_list(box, src=0)
                    return box[0]
            ident = exchange_id(ident if rank == 0 else None)
            if not isinstance(ident, (bytes, bytearray)) or len(ident) != 128:
                raise cabi.LadeHipError("exchange_id must return rank 0's 128-byte unique id on every rank")
        handle = C.c_void_p()
        cabi.call_plain("lade_lp_comm_create", C.create_string_buffer(bytes(ident), 128), rank, world, C.byref(handle))
        self.handle = handle

    def count(self) -> int:
        """ranks the communicator spans (ncclCommCount)"""
        import ctypes as C
        n = C.c_int32(0)
        self._cabi.call_plain("lade_lp_comm_count", self.handle, C.byref(n))
        return int(n.value)

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        assert out.dtype == torch.int32 and inp.dtype == torch.int32 and out.numel() == self.world * inp.numel()
        self._cabi.call("lade_lp_allgather", self.handle, self._cabi.ptr(inp), self._cabi.ptr(out), inp.numel())

    def close(self) -> None:
        if self.handle is not None:
            self._cabi.call_plain("lade_lp_comm_destroy", self.handle)
            self.handle = None


# ---- rank 0's GEMM autotune table as a fixed int32 block (so that it can travel through the step's own all-gather) ----
_TUNE_FIELDS = 7           # present flag + (mb, bn, S, mt, nt, ring)


def encode_tune_table(table: dict, names: Sequence[str], classes: Sequence[int]) -> List[int]:
    out: List[int] = []
    for n in names:
        for m in classes:
            v = table.get(f"{n}:{m}")
            out += [0] * _TUNE_FIELDS if v is None else [1] + ([int(x) for x in v] + [0])[:_TUNE_FIELDS - 1]       # (5-tuples of older tables: default ring)
    return out


def decode_tune_table(words: Sequence[int], names: Sequence[str], classes: Sequence[int]) -> dict:
    out, i = {}, 0
    for n in names:
        for m in classes:
            w = [int(x) for x in words[i:i + _TUNE_FIELDS]]
            out[f"{n}:{m}"] = tuple(w[1:]) if w[0] else None
            i += _TUNE_FIELDS
    return out


class HipLPBackend:
    """Rank-local device work of one LP step on the HIP kernels (wraps a LookaheadDecoder's engine + state)."""

    def __init__(self, dec):
        from . import ops
        from .cabi import call, ptr
        self.dec, self.ops, self.call, self.ptr = dec, ops, call, ptr
        st = dec.st
        self.rw = rec_words(dec.gs, st.wcap, dec.G)
        dev = dec.e.device
        self.rec = torch.zeros(self.rw, dtype=torch.int32, device=dev)
        self.scratch = torch.zeros(dec.lp.R * st.wcap + dec.G * dec.gs + 64, dtype=torch.int32, device=dev)
        self.device = dev

    def begin(self, prompt: Sequence[int], window0: Sequence[int], eos: int) -> None:
        d = self.dec
        d.e.reset()
        d.st.reset(window0, len(prompt), prompt)
        self.prompt = list(prompt)
        self.window0 = list(window0)
        self.eos = eos
        if d.pool_from_prompt:                                   # lade/decoding.py:915-916, on every rank
            pt = torch.tensor(self.prompt, dtype=torch.int32, device=self.device)
            self.call("lade_pool_fill_prompt", self.ptr(d.st.pool_tok), self.ptr(d.st.pool_cnt), d.st.V, d.G, d.gs, self.ptr(pt), len(self.prompt))

    def local_step(self, phase: int, P: int, n_input: int, level_lens: Sequence[int], c0: int, c1: int, g: int, glo: int, ghi: int):
        """Builds this rank's inputs, runs the forward, and packs its record (device tensor [rw])."""
        from .ops import StepMask
        d, st, e = self.dec, self.dec.st, self.dec.e
        gs, N = d.gs, d.N
        call, ptr = self.call, self.ptr
        g_local = ghi - glo
        cand_rows = g_local * gs if phase == 2 else 0
        if phase == 0:
            # rank r prefills the prompt + the L0 prefix up to its last column (lade/decoding.py:981-984)
            ids_h = self.prompt + self.window0[: c1 - 1]
            n_inp = c1 - 1
            logits, done = e.prefill(ids_h, [len(self.prompt) - 1] + list(range(len(self.prompt), len(ids_h))))
        else:
            ls = shard_level_sizes(level_lens, c0, c1)
            mask = StepMask.from_levels(n_input, ls, cand_rows, gs, P)
            T = mask.T
            guess_ptr = st.guess.data_ptr() + 4 * glo * gs
            call("lade_build_inputs", None, None, n_input, ptr(st.window), st.wcap, ptr(st.ctl), len(level_lens) - 1, c0, c1,
                 guess_ptr, g_local if phase == 2 else 0, gs, cand_rows, ptr(st.ids), ptr(st.pos), None, 0, 1)
            n_inp = ls[-1]
            rows = [n_input - 1] + list(range(T - cand_rows - n_inp, T - cand_rows)) + list(range(T - cand_rows, T))
            d._set_sel(rows)                 # cached on the device per shape: no host-to-device copy in a steady step
            logits = e.forward(st.ids, st.pos, mask, st.sel, len(rows), argmax_out=st.am)
        if logits is not None:
            self.ops.argmax_rows(logits, out=st.am)
        am = st.am.data_ptr()
        call("lade_lp_pack", am, am + 4, n_inp, am + 4 * (1 + n_inp), g_local if phase == 2 else 0, gs, st.wcap, ptr(self.rec), self.rw, None, 0, 1)
        return self.rec

    # ---- steady step as a hipGraph segment: [input assembly -> forward -> argmax -> record] ---------------------------------------
    # Fixed shapes per (re-fed input count, candidate bucket): the rank's candidate share is decided on the device from ctl[G]
    # (lade_build_inputs / lade_lp_pack with lp_rank, lp_world), unused candidate slots are padded, the cache length is read by the
    # kernels from ctl[0].  The collective and lade_lp_reduce_apply follow on the same stream; the host only picks the segment and
    # reads the step's record.
    _capture_lock = __import__("threading").Lock()

    def _local_buckets(self) -> List[int]:
        gl = (self.dec.G + self.dec.lp.R - 1) // self.dec.lp.R
        return sorted({0, (gl + 1) // 2, gl})

    def _segment_body(self, n_input: int, ls: Sequence[int], c0: int, c1: int, gcap: int, sel: torch.Tensor, n_splits: int):
        from .ops import StepMask
        d, st, e = self.dec, self.dec.st, self.dec.e
        gs, lp = d.gs, d.lp
        call, ptr = self.call, self.ptr
        cand_rows = gcap * gs
        mask = StepMask.from_levels(n_input, ls, cand_rows, gs, 0)
        call("lade_build_inputs", None, None, n_input, ptr(st.window), st.wcap, ptr(st.ctl), d.N - 2, c0, c1, ptr(st.guess), -1, gs, cand_rows,
             ptr(st.ids), ptr(st.pos), None, lp.rank, lp.R)
        e.forward(st.ids, st.pos, mask, sel, sel.numel(), dyn_P=st.ctl, n_splits=n_splits, argmax_out=st.am)
        am = st.am.data_ptr()
        n_inp = ls[-1]
        call("lade_lp_pack", am, am + 4, n_inp, am + 4 * (1 + n_inp), gcap, gs, st.wcap, ptr(self.rec), self.rw, ptr(st.ctl), lp.rank, lp.R)

    def _capture_segment(self, P: int, n_input: int, ls: Sequence[int], c0: int, c1: int, gcap: int):
        d, st, e = self.dec, self.dec.st, self.dec.e
        gs = d.gs
        T = n_input + sum(ls) + gcap * gs
        rows = [n_input - 1] + list(range(T - gcap * gs - ls[-1], T))
        sel = torch.tensor(rows, dtype=torch.int32, device=self.device)
        n_splits = e.n_splits_for(T, max(P + T, 1024))
        state = (st.ctl, st.window, st.pool_cnt, st.pool_tok, st.guess, st.tail)
        with HipLPBackend._capture_lock:           # one capture at a time per process (ranks may be threads in the tests)
            saved = [t.clone() for t in state]
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    self._segment_body(n_input, ls, c0, c1, gcap, sel, n_splits)       # warm-up: library handles, autotune
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
            finally:                                   # whatever the warm-up did to the integer state is undone - also when it failed (LPRunner falls back to eager steps)
                for t, sv in zip(state, saved):
                    t.copy_(sv)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._segment_body(n_input, ls, c0, c1, gcap, sel, n_splits)
        return (g, sel, n_splits)

    def local_step_graph(self, P: int, n_input: int, level_lens: Sequence[int], c0: int, c1: int, g_local: int):
        """Replays (capturing on first use) the segment for this step's shape; returns the record tensor.  The first steady step
        (one input token) captures the segments of ALL candidate buckets at once, like the single-GPU decoder does: the GEMM autotune
        of a wider row class then happens during warm-up, not in the step that first meets a candidate."""
        d, st, e = self.dec, self.dec.st, self.dec.e
        gs = d.gs
        ls = shard_level_sizes(level_lens, c0, c1)
        gcap = min(b for b in self._local_buckets() if b >= g_local)
        T = n_input + sum(ls) + gcap * gs
        if P + T > e.S_max:
            raise self.dec_error(f"KV cache exhausted: P={P} + T={T} > S_max={e.S_max}")
        key = (n_input, gcap, e.generation)
        graphs = self.__dict__.setdefault("_segments", {})
        want_splits = e.n_splits_for(T, P + T)
        ent = graphs.get(key)
        if ent is None or abs(want_splits - ent[2]) >= 2:
            ent = graphs[key] = self._capture_segment(P, n_input, ls, c0, c1, gcap)
            if n_input == 1:
                for b in self._local_buckets():
                    if (1, b, e.generation) not in graphs and P + 1 + sum(ls) + b * gs <= e.S_max:
                        graphs[(1, b, e.generation)] = self._capture_segment(P, 1, ls, c0, c1, b)
        ent[0].replay()
        return self.rec

    @staticmethod
    def dec_error(msg):
        from .cabi import LadeHipError
        return LadeHipError(msg)

    def apply(self, all_rec: torch.Tensor, R: int, phase: int) -> List[int]:
        d, st = self.dec, self.dec.st
        call, ptr = self.call, self.ptr
        call("lade_lp_reduce_apply", ptr(all_rec), R, self.rw, st.wcap, ptr(st.ctl), ptr(st.window), st.wcap, ptr(st.pool_tok), ptr(st.pool_cnt),
             st.V, d.W, d.N, d.G, phase, ptr(st.guess), ptr(self.scratch), ptr(st.record), int(d.pool_from_prompt), ptr(st.tail), self.eos)
        return st.read_record()

    def new_gather_buffer(self, R: int) -> torch.Tensor:
        return torch.zeros(R * self.rw, dtype=torch.int32, device=self.device)

    comm = None          # RcclComm when the step's collective goes through the C ABI (set by LPRunner)

    def broadcast_window(self, window0: List[int], lp: LPContext) -> List[int]:
        """rank 0's random window reaches every rank (lade/decoding.py:905-906)"""
        if lp.R == 1:
            return list(window0)
        t = torch.tensor(window0, dtype=torch.int32, device=self.device)
        if self.comm is not None:                       # no torch.distributed in play: gather everybody's window, keep rank 0's
            allw = torch.zeros(lp.R * t.numel(), dtype=torch.int32, device=self.device)
            self.comm.all_gather(allw, t)
            t = allw[:t.numel()]
        else:
            dist.broadcast(t, src=0, group=lp.group)
        self.sync_gemm_choice(lp)
        return t.tolist()

    def sync_gemm_choice(self, lp: LPContext) -> None:
        """Once per engine: rank 0's autotuned GEMM table is adopted by every rank (each process would otherwise time its own
        candidates and could settle on kernels that round 16-bit results differently).  With the C ABI's communicator the table
        travels as a fixed int32 block through the same all-gather the step uses - no torch.distributed needed; otherwise through
        the process group's object broadcast.  A world of more than one rank with neither is refused rather than left to race."""
        e = self.dec.e
        if getattr(e, "_lp_gemm_synced", False) or not e.custom_gemm or lp.R == 1:
            return
        if self.comm is not None:
            words = encode_tune_table(e.tune_all() if lp.rank == 0 else {}, e.TUNE_NAMES, e.ROW_CLASSES)
            mine = torch.tensor(words, dtype=torch.int32, device=self.device)
            allw = torch.zeros(lp.R * mine.numel(), dtype=torch.int32, device=self.device)
            self.comm.all_gather(allw, mine)
            if lp.rank != 0:
                e.adopt_gemm_cfg(decode_tune_table(allw[:mine.numel()].tolist(), e.TUNE_NAMES, e.ROW_CLASSES))
        elif dist.is_available() and dist.is_initialized():
            box = [e.tune_all() if lp.rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=lp.group)
            if lp.rank != 0:
                e.adopt_gemm_cfg(box[0])
        else:
            raise self.dec_error("lookahead parallelism with more than one rank needs a channel for rank 0's GEMM table: the C ABI "
                                 "communicator (LADE_LP_COLLECTIVE=abi / LPRunner(comm=...)) or an initialised torch.distributed group")
        e._lp_gemm_synced = True


class LPRunner:
    """Stepwise driver of greedy lookahead decoding under lookahead parallelism (start / step), used by
    `greedy_lp` and by bench.py.  `dec` supplies W, N, G and the LPContext; `backend` defaults to the HIP backend."""

    def __init__(self, dec, backend=None, all_gather=None, comm: Optional[RcclComm] = None):
        self.dec = dec
        self.lp: LPContext = dec.lp
        if comm is not None:                 # a communicator the caller built over its own channel (no torch.distributed involved)
            self.comm = comm
            all_gather = comm.all_gather
        # the collective is injectable so that tests can run several ranks inside one process.  Default: torch.distributed's
        # all_gather_into_tensor (RCCL); LADE_LP_COLLECTIVE=abi routes it through the C ABI's own communicator
        # (lade_lp_allgather) instead - the path a caller without torch.distributed uses.
        if all_gather is None:
            if os.environ.get("LADE_LP_COLLECTIVE", "torch") == "abi":
                self.comm = RcclComm(self.lp.rank, self.lp.R)
                all_gather = self.comm.all_gather
            elif self.lp.R == 1 and not (dist.is_available() and dist.is_initialized()):
                # one rank and no process group (CONFIG_MAP["FORCE_LP"] on a single GPU; the reference joins no group when
                # DIST_WORKERS == 1, lade/utils.py:28-33): the gather of one record is a device copy on the step's stream - which is
                # also all RCCL does for one rank (profiles/r3_lp_collective_kernels.txt)
                all_gather = lambda out, inp: out.copy_(inp)
            else:
                all_gather = lambda out, inp: dist.all_gather_into_tensor(out, inp, group=self.lp.group)
        self.all_gather = all_gather
        self.W, self.N, self.G = dec.W, dec.N, dec.G
        if self.lp.R > self.W:
            raise ValueError(f"lookahead parallelism needs DIST_WORKERS ({self.lp.R}) <= WINDOW_SIZE ({self.W})")
        self.be = backend if backend is not None else HipLPBackend(dec)
        if getattr(self, "comm", None) is not None:
            self.be.comm = self.comm

    def start(self, prompt: Sequence[int], eos_token_id: Optional[int] = None, rng: Optional[random.Random] = None) -> None:
        rng = rng if rng is not None else random
        W, N = self.W, self.N
        self.prompt = [int(t) for t in prompt]
        # every rank draws its own window, rank 0's is broadcast (lade/decoding.py:902-906)
        window0 = [rng.choice(self.prompt) for _ in range(W + N - 3)]
        window0 = self.be.broadcast_window(window0, self.lp)
        self.eos = -1 if eos_token_id is None else int(eos_token_id)
        self.be.begin(self.prompt, window0, self.eos)
        self.all_rec = self.be.new_gather_buffer(self.lp.R)
        self.tokens = list(self.prompt)
        self.steps, self.P, self.g, self.fill_level, self.n_input = 0, 0, 0, 0, len(self.prompt)
        self.finished = False

    def step(self) -> dict:
        W, N, R, r = self.W, self.N, self.lp.R, self.lp.rank
        fill_level = self.fill_level
        phase = 0 if self.steps == 0 else (2 if fill_level >= N - 2 else 1)
        # level lengths before this step (see LookaheadDecoder._level_sizes)
        if fill_level == 0:
            level_lens = [W + N - 3]
        elif fill_level >= N - 2:
            level_lens = [W - 1] + [W] * (N - 2)
        else:
            level_lens = [W + N - 3 - fill_level] + [W + N - 2 - fill_level] * fill_level
        c0, c1 = window_shard(level_lens[0] + 1, R, r)
        glo, ghi = guess_shard(self.g, R, r) if phase == 2 else (0, 0)
        # (dynamic-NTK RoPE: a padded graph segment would count its padding as sequence length - those models take the eager LP step)
        if phase == 2 and getattr(self.dec, "use_graph", False) and hasattr(self.be, "local_step_graph") and getattr(self.dec.e, "ntk_state", None) is None:
            try:
                rec = self.be.local_step_graph(self.P, self.n_input, level_lens, c0, c1, ghi - glo)
            except (RuntimeError, OSError) as ex:
                if type(ex).__name__ == "LadeHipError":
                    raise                                 # a refusal of the step itself (cache exhausted ...), not of the capture
                # the hipGraph segment could not be captured / replayed on this box (never exercised next to a multi-GPU RCCL communicator
                # before round 5): this rank takes the eager step from here on - the other ranks need not, the step's collective is the same
                import sys
                print(f"[lade] rank {r}: hipGraph segment unavailable ({type(ex).__name__}: {str(ex)[:200]}); eager lookahead-parallel steps from here on",
                      file=sys.stderr, flush=True)
                self.dec.use_graph = False
                rec = self.be.local_step(phase, self.P, self.n_input, level_lens, c0, c1, self.g, glo, ghi)
        else:
            rec = self.be.local_step(phase, self.P, self.n_input, level_lens, c0, c1, self.g, glo, ghi)
        self.all_gather(self.all_rec, rec)                                       # the ONE exchange of the step
        out = self.be.apply(self.all_rec, R, phase)
        self.steps += 1
        max_hit, n_accept, self.g, self.P = out[0], out[1], out[3], out[4]
        accepted = out[8:8 + n_accept]
        # EOS scan (lade/decoding.py:1167-1177): the reference keeps everything up to and including EOS
        if self.eos >= 0 and self.eos in accepted:
            accepted = accepted[:accepted.index(self.eos) + 1]
            self.finished = True
        self.tokens += accepted
        self.n_input = 1 + max_hit                                           # re-feed the hits (:1148-1153)
        if phase != 2:
            self.fill_level += 1
        return dict(phase=phase, max_hit=max_hit, accepted=list(accepted), c0=c0, c1=c1, glo=glo, ghi=ghi, P_after=self.P, g_next=self.g,
                    T=None)


def greedy_lp(dec, prompt: Sequence[int], max_length: int, eos_token_id: Optional[int] = None,
              rng: Optional[random.Random] = None, keep_trace: bool = False, backend=None, all_gather=None, on_step=None, comm=None):
    """Greedy lookahead decoding under lookahead parallelism (jacobi_greedy_search_multilevel with
    DIST_WORKERS > 1, lade/decoding.py:697-1259)."""
    from .decoding import GenOut
    run = LPRunner(dec, backend, all_gather, comm=comm)
    run.start(prompt, eos_token_id, rng)
    trace: List[dict] = []
    while True:
        info = run.step()
        if keep_trace:
            trace.append(info)
        if on_step is not None:
            on_step(info["accepted"][:max(0, max_length - (len(run.tokens) - len(info["accepted"])))])
        if run.finished or len(run.tokens) >= max_length:
            break
    generated = min(len(run.tokens), max_length) - len(run.prompt)
    return GenOut(tokens=run.tokens[:max_length], steps=run.steps, generated=generated, trace=trace)
