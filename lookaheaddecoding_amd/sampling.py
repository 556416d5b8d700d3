"""Host side of the sampling path: logits warpers and the multi-candidate rejection-sampling verify.

Reference: lade/decoding.py:375-377 (admitted warpers: Temperature, TopK, TopP - applied through HF's
LogitsProcessorList at :443, :488) and :484-540 (verification, "modified from specinfer").  The verify loop
stays on the host on purpose: it draws `random.random()` once per trial and `torch.multinomial` once per
sampled token, and the order of those draws is part of the behaviour to reproduce; the probabilities it
reads are computed on the GPU and fetched row by row only when a candidate prefix is accepted.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch


def make_warper(temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0) -> Callable[[torch.Tensor], torch.Tensor]:
    """Temperature -> top-k -> top-p on [rows, V] fp32 logits (same order and tie rules as HF's
    TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper); runs on whatever device the logits are on."""
    def warp(x: torch.Tensor) -> torch.Tensor:
        if temperature != 1.0:
            x = x / temperature
        if top_k and top_k > 0:
            k = min(top_k, x.shape[-1])
            kth = torch.topk(x, k)[0][..., -1, None]
            x = x.masked_fill(x < kth, -float("inf"))
        if top_p < 1.0:
            sl, si = torch.sort(x, descending=False)
            cp = sl.softmax(dim=-1).cumsum(dim=-1)
            rm = cp <= (1 - top_p)
            rm[..., -1:] = False
            x = x.masked_fill(rm.scatter(-1, si, rm), -float("inf"))
        return x
    return warp


def sample_verify(probs_next: torch.Tensor, guess_probs_row: Callable[[int], torch.Tensor], guess_tokens: Sequence[int], gs: int,
                  rng, multinomial: Callable[[torch.Tensor], int]) -> Tuple[List[int], int]:
    """lade/decoding.py:484-540.  probs_next: CPU fp32 [V] (consumed); guess_probs_row(r) returns the CPU
    probabilities that follow candidate row r.  Position k of the n-gram: walk the surviving candidates in
    order, accept draft d with probability min(1, probs_next[d]) (one rng.random() per trial); on accept keep
    the candidates that share d and continue from that row's distribution; on reject zero d and renormalise;
    when every survivor is rejected, sample from what is left and stop.  At most gs tokens."""
    probs_next = probs_next.clone()
    hits: List[int] = []
    n_cand = len(guess_tokens) // gs
    guess_indices = list(range(n_cand))
    max_hit_idx = 0
    for idx_in_ngram in range(gs):
        g_idx, is_accept = 0, False
        guess_offset = 0
        while g_idx < len(guess_indices):
            guess_idx = guess_indices[g_idx]
            guess_offset = guess_idx * gs
            draft = guess_tokens[guess_offset + idx_in_ngram]
            prob_accept = min(1, probs_next[draft].item())
            if rng.random() < prob_accept:
                hits.append(draft)
                is_accept = True
                max_hit_idx = guess_idx
                guess_indices = [gi for gi in guess_indices if guess_tokens[gi * gs + idx_in_ngram] == draft]
                break
            probs_next[draft] = 0
            probs_next = probs_next / probs_next.sum()
            g_idx += 1
        if is_accept:
            probs_next = guess_probs_row(guess_offset + idx_in_ngram).clone()
            continue
        hits.append(int(multinomial(probs_next)))
        break
    return hits, max_hit_idx
