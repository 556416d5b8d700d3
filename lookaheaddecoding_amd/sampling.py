"""Sampling path: logits warpers and the host half of the multi-candidate rejection-sampling verify.

Reference: lade/decoding.py:375-377 (admitted warpers: Temperature, TopK, TopP, applied through HF's LogitsProcessorList at
:443, :488) and :484-540 (the verify loop, "modified from specinfer").

Split of the work:

* device (`lade_softmax_gather`): softmax statistics of the out row and of every candidate row, and - the only numbers the
  acceptance loop ever looks at - the probability of each candidate's draft token under the distribution in force.  The
  reference materialises `guess_probs` ([g*gs, V] fp32, 7.7 MB per step at config 3) and indexes single elements of it; here
  the step hands over one small [1 + g*gs, g] table.
* host (`resolve_drafts`): the draws.  The order in which `random.random()` is consumed (one per trial, candidates in pool
  order, position by position) is part of the behaviour to reproduce, so the walk over trials stays on the host; it reads
  scalars only.  A rejected draft is struck from the distribution and the rest renormalised - for the trials that is one
  running scale factor.
* the token that ends the step is drawn from ONE full distribution row (`final_distribution`): the row in force with the
  struck drafts removed, computed on the device (`lade_softmax_rows`), renormalised strike by strike like the reference does.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import torch


class Warper:
    """Temperature -> top-k -> top-p on [rows, V] fp32 logits (same order and tie rules as HF's TemperatureLogitsWarper /
    TopKLogitsWarper / TopPLogitsWarper).  With top-k and top-p off the warp is a plain scale, which the device kernels apply
    themselves (`fused_temperature`); otherwise the sampling loop calls `warp_rows` - one HIP launch (lade_warp_rows: no sort, no
    topk; beyond 32768 tokens its output-row form) - and `__call__` (torch ops on whatever device the logits are on) remains for CPU
    tensors and as the plain statement of the semantics."""

    def __init__(self, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0):
        self.temperature, self.top_k, self.top_p = float(temperature), int(top_k or 0), float(top_p)

    @property
    def fused_temperature(self) -> Optional[float]:
        return self.temperature if (self.top_k <= 0 and self.top_p >= 1.0) else None

    def warp_rows(self, logits: torch.Tensor, rows: int, skip: int) -> Optional[torch.Tensor]:
        """The warp of logical rows 0 and 1 + skip .. (the out row and the candidate rows of a step's logits) as ONE device launch
        (lade_warp_rows); None for a CPU tensor or a vocabulary beyond 2^24 - the caller then uses __call__ (torch ops)."""
        from . import ops
        if not logits.is_cuda or logits.shape[-1] > ops.WARP_MAX_V:
            return None
        return ops.warp_rows(logits, rows, skip, self.temperature, self.top_k, self.top_p)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if self.temperature != 1.0:
            x = x / self.temperature
        if self.top_k > 0:
            k = min(self.top_k, x.shape[-1])
            kth = torch.topk(x, k)[0][..., -1, None]
            x = x.masked_fill(x < kth, -float("inf"))
        if self.top_p < 1.0:
            sl, si = torch.sort(x, descending=False)
            cp = sl.softmax(dim=-1).cumsum(dim=-1)
            rm = cp <= (1 - self.top_p)
            rm[..., -1:] = False
            x = x.masked_fill(rm.scatter(-1, si, rm), -float("inf"))
        return x


def make_warper(temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0) -> Warper:
    return Warper(temperature, top_k, top_p)


@dataclass
class Verdict:
    """Outcome of the acceptance walk of one step."""
    accepted: List[int]                  # drafts accepted, in order (at most gs)
    winner: int                          # candidate whose draft was accepted last (max_hit_idx; 0 when none)
    final_row: Optional[int]             # logical row (0 = out row, 1 + c*gs + j) a token must still be drawn from; None when all
                                         # gs positions were accepted (the walk ends without a draw)
    struck: List[int] = field(default_factory=list)     # drafts rejected under final_row, in the order they were struck


def resolve_drafts(table: Sequence[Sequence[float]], drafts: Sequence[int], g: int, gs: int, uniform: Callable[[], float]) -> Verdict:
    """The acceptance walk of lade/decoding.py:491-534 over the device-gathered table.

    table[row][c] = P(draft of candidate c at the position `row` judges | distribution `row`); row 0 judges position 0, row
    1 + c'*gs + j judges position j + 1 (what follows the prefix candidate c' shares up to j).  drafts = the g candidates'
    tokens, candidate-major.  Position by position: the candidates that still match the accepted prefix are tried in pool order;
    a draft of current probability p is accepted when uniform() < min(1, p).  A rejected draft is removed from the distribution
    and the remainder renormalised: every other probability grows by 1 / (1 - p), a draft that was already removed has p = 0.
    One uniform() per trial, whatever its outcome."""
    alive = list(range(g))
    row, accepted, winner = 0, [], 0
    for k in range(gs):
        grow, struck, hit = 1.0, [], None
        for c in alive:
            d = drafts[c * gs + k]
            p = 0.0 if d in struck else float(table[row][c]) * grow
            if uniform() < min(1.0, p):
                hit = (c, d)
                break
            if d not in struck:
                struck.append(d)
                grow = grow / (1.0 - p) if p < 1.0 else grow
        if hit is None:
            return Verdict(accepted, winner, row, struck)
        winner = hit[0]
        accepted.append(hit[1])
        alive = [c for c in alive if drafts[c * gs + k] == hit[1]]
        row = 1 + winner * gs + k
    return Verdict(accepted, winner, None, [])


def multinomial_one(probs_row: torch.Tensor, generator: Optional[torch.Generator]) -> torch.Tensor:
    """`torch.multinomial(probs_row, num_samples=1, generator=generator)` (the draw of lade/decoding.py:484-540) for a device row,
    launch for launch what torch runs behind its input checks: q ~ Exp(1) from the generator, argmax(p / q) - the SAME token from the
    same generator state (tests/test_gpu_kernels.py draws both ways from cloned states), without the twelve launches that validate the
    row first (max < inf, min >= 0, sum > 0: two device-side asserts) - the row comes from `lade_softmax_rows`, which cannot produce
    what they reject unless the logits were not finite.  Returns the index as an int64 tensor [1] on the row's device."""
    q = torch.empty_like(probs_row).exponential_(1.0, generator=generator)
    torch.div(probs_row, q, out=q)
    return torch.argmax(q, dim=-1, keepdim=True)


def final_distribution(probs_row: torch.Tensor, struck: Sequence[int]) -> torch.Tensor:
    """probs_row [V] fp32 (any device, consumed): the drafts rejected under this row are removed one at a time, renormalising
    after each (lade/decoding.py:519-520) - the arithmetic of the reference, on the one row that is actually sampled from."""
    for d in struck:
        probs_row[d] = 0
        probs_row = probs_row / probs_row.sum()
    return probs_row
