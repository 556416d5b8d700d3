"""CPU oracle for the lookahead-decoding step  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch CPU restatement (python ints / numpy / torch-CPU fp32) of the reference's
per-step hot path, used ONLY as the checker by `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg.  Nothing under `lookaheaddecoding_amd/` imports this file;
the product path is the HIP extension and fails loudly without it.

Parity status: PINNED.  Every function below is checked in `tests/test_oracle_golden.py`
against fixtures under `tests/golden/` that were produced by running the reference's OWN code
(`/root/reference/lade`, unmodified, through `oracle/ref_shim.py`) in the build container with
`oracle/make_golden.py` (pool KATs, dense masks, step input assembly, attention, and whole
greedy / sampling / lookahead-parallel token+step traces).

Each function cites the reference lines it follows (paths relative to /root/reference).
Symbols: W=WINDOW_SIZE, N=LEVEL, G=GUESS_SET_SIZE, gs=N-1 (tokens per candidate),
g=#candidates verified this step, P=KV rows before the step, T=tokens fed this step.
"""
from __future__ import annotations

import math
import random as _random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------------------
# 1. n-gram pool  (lade/decoding.py:37-127)
# --------------------------------------------------------------------------------------

def _lru_insert(token_map: dict, key: int, tup: tuple, G: int) -> None:
    """One LRU insert of `tup` under `key` (lade/decoding.py:39-49, repeated :52-63, :88-96, :114-122).

    present -> move to the end; room -> append; full -> drop the head, append."""
    lst = token_map.setdefault(key, [])
    if tup in lst:
        lst.remove(tup)
        lst.append(tup)
    elif len(lst) < G:
        lst.append(tup)
    else:
        assert len(lst) == G
        token_map[key] = lst[1:] + [tup]


def update_token_map(token_map, lst_token, past_tokens, new_results, LEVEL, WINDOW_SIZE, GUESS_SET_SIZE):
    """W sequential n-gram inserts per steady step (lade/decoding.py:37-78).

    column i: key = lst_token (i=0) or past_tokens[0][i-1]; value = (past_tokens[1][i], ...,
    past_tokens[N-2][i], new_results[i]).  GUESS_SET_SIZE == -1 -> python `set` per key."""
    for i in range(WINDOW_SIZE):
        key = lst_token if i == 0 else past_tokens[0][i - 1]
        tup = tuple(past_tokens[ll][i] for ll in range(1, LEVEL - 1)) + (new_results[i],)
        if GUESS_SET_SIZE != -1:
            _lru_insert(token_map, key, tup, GUESS_SET_SIZE)
        else:
            token_map.setdefault(key, set()).add(tup)


def append_new_generated_pool(tokens, token_map, LEVEL, GUESS_SET_SIZE):
    """lade/decoding.py:80-101 -- one insert of the last N accepted tokens (POOL_FROM_PROMPT)."""
    if len(tokens) != LEVEL:
        return
    key, tup = tokens[0], tuple(tokens[1:])
    if GUESS_SET_SIZE != -1:
        _lru_insert(token_map, key, tup, GUESS_SET_SIZE)
    else:
        token_map.setdefault(key, set()).add(tup)


def fill_pool_with_prompt(prompts, token_map, LEVEL, GUESS_SET_SIZE):
    """lade/decoding.py:104-127 -- sliding N-grams over the prompt."""
    for start in range(len(prompts) - LEVEL + 1):
        key = prompts[start]
        tup = tuple(prompts[start + 1:start + LEVEL])
        if len(tup) != LEVEL - 1:
            return
        if GUESS_SET_SIZE != -1:
            _lru_insert(token_map, key, tup, GUESS_SET_SIZE)
        else:
            token_map.setdefault(key, set()).add(tup)


def filter_window(level_window, eos_token_id, reset_func):
    """lade/decoding.py:131-135 -- sampling path only."""
    for idx in range(len(level_window)):
        if level_window[idx] == eos_token_id:
            level_window[idx] = reset_func()


def pool_lookup(token_map, lst_token, window_full: bool, GUESS_SET_SIZE: int) -> Optional[List[int]]:
    """lade/decoding.py:948-954 (greedy) / :402-406 (sample): flatten <=G tuples in list order."""
    if window_full and lst_token in token_map and GUESS_SET_SIZE > 0:
        out: List[int] = []
        for tok in list(token_map[lst_token]):
            out += list(tok)
        return out
    return None

# --------------------------------------------------------------------------------------
# 2. greedy verify  (lade/decoding.py:1071-1084)
# --------------------------------------------------------------------------------------

def greedy_verify(first_guess: int, guess_tokens: Optional[Sequence[int]], guess_results: Sequence[int], gs: int):
    """Longest-prefix match over candidates.  Returns (max_hit, max_hit_idx, hits[gs]).

    For candidate e: correct = [first_guess] + argmax ids of its gs rows; gg = first index
    where the candidate differs from `correct`, or gs-1 when none differ (the python loop
    variable simply ends at gs-1: a full match drops the bonus token).
    Only a strictly greater gg replaces the current best (first candidate wins ties)."""
    max_hit, max_hit_idx = 0, 0
    hits = [first_guess] + [0] * (gs - 1)
    if guess_tokens is not None:
        for eg in range(len(guess_results) // gs):
            egx = eg * gs
            correct = [first_guess] + list(guess_results[egx:egx + gs])
            myguess = guess_tokens[egx:egx + gs]
            gg = 0
            for gg in range(len(myguess)):
                if myguess[gg] != correct[gg]:
                    break
            if gg > max_hit:
                max_hit, max_hit_idx = gg, eg
                hits[:max_hit + 1] = correct[:max_hit + 1]
    return max_hit, max_hit_idx, hits

# --------------------------------------------------------------------------------------
# 3. Jacobi window fill + roll  (lade/decoding.py:1038-1066, 1119-1124)
# --------------------------------------------------------------------------------------

def window_fill_first(past_tokens, inp_argmax):
    """Step 1 (prefill) -- lade/decoding.py:1038-1048."""
    past_tokens[0] = past_tokens[0][1:]
    past_tokens[1] = list(inp_argmax)


def window_fill(past_tokens, fill_level, inp_argmax_all):
    """Fill step -- lade/decoding.py:1049-1066: drop the head of levels 0..fill_level, the new
    level is argmax(inp_logits)[1:]."""
    for level in range(fill_level + 1):
        past_tokens[level] = past_tokens[level][1:]
    past_tokens[fill_level + 1] = list(inp_argmax_all)[1:]


def window_roll(past_tokens, new_results, LEVEL):
    """Steady roll (ALWAYS_FWD_ONE=1) -- lade/decoding.py:1119-1124."""
    past_tokens[0] = past_tokens[1][1:]
    for level in range(1, LEVEL - 2):
        past_tokens[level] = past_tokens[level + 1][:]
    past_tokens[LEVEL - 2] = list(new_results)

# --------------------------------------------------------------------------------------
# 4. lookahead-parallel partitioning  (lade/decoding.py:956-963, 973-986)
# --------------------------------------------------------------------------------------

def lp_window_shard(past_tokens, R: int, r: int):
    """Rank r's slice of the window: the whole L0 prefix up to its last column, and its own
    columns of every higher level (lade/decoding.py:973-984).  Returns (past_tokens_inp,
    window_start, window_end)."""
    window_len = len(past_tokens[0]) + 1
    split = (window_len + R - 1) // R
    ws = min(split * r, window_len)
    we = min(split * (r + 1), window_len)
    inp = [past_tokens[0][: we - 1]]
    for l in range(1, len(past_tokens)):
        toks = past_tokens[l]
        inp.append(toks[ws:we] if toks is not None else None)
    return inp, ws, we


def lp_guess_shard(guess_tokens: Optional[List[int]], gs: int, R: int, r: int):
    """Rank r's candidates (lade/decoding.py:956-963)."""
    if guess_tokens is None:
        return None
    assert len(guess_tokens) % gs == 0
    cnt = (len(guess_tokens) // gs + R - 1) // R
    out = guess_tokens[gs * cnt * r: gs * cnt * (r + 1)]
    return out if len(out) > 0 else None

# --------------------------------------------------------------------------------------
# 5. step input assembly  (lade/models/modeling_llama.py:1463-1513) and mask parameters
# --------------------------------------------------------------------------------------

@dataclass
class StepLayout:
    """Everything that describes one forward of `jforward_multilevel` (eager token order)."""
    ids: List[int]
    positions: List[int]
    n_input: int                 # 1 + guess_skip_dist (or the prompt length at prefill)
    level_sizes: List[int]
    lguess: int                  # g * gs
    is_prefill: bool
    window: int                  # rows of inp_logits = len(past_tokens[fill_level])

    @property
    def T(self) -> int:
        return len(self.ids)

    @property
    def level_offset(self) -> int:  # modeling_llama.py:137 / :1181
        return self.T - (sum(self.level_sizes) + 1) - self.lguess

    @property
    def dist_offset(self) -> int:   # modeling_llama.py:138 / :1182
        return 1 + self.level_sizes[0] - self.level_sizes[-1]


def build_step_layout(input_ids: Sequence[int], position_ids: Sequence[int], past_tokens, guess_tokens,
                      fill_level: int, gs: int) -> StepLayout:
    """Token order [inputs | L0 | L1 | ... | L_fill | candidates]; positions: L0 -> lst_id+1+i;
    level l>=1 column i -> lst_id + l + (len(L0)+1-len(Ll)) + i; candidates restart at lst_id+1
    (lade/models/modeling_llama.py:1487-1511)."""
    lst_id = position_ids[-1]
    all_past: List[int] = []
    ids_list: List[int] = []
    level_sizes: List[int] = []
    for ll in range(fill_level + 1):
        all_past += past_tokens[ll]
        level_sizes.append(len(past_tokens[ll]))
        if ll == 0:
            ids_list += list(range(lst_id + 1, lst_id + 1 + len(past_tokens[ll])))
        else:
            off = len(past_tokens[0]) + 1 - len(past_tokens[ll])
            ids_list += list(range(lst_id + ll + off, lst_id + ll + off + len(past_tokens[ll])))
    ids = list(input_ids) + all_past
    pos = list(position_ids) + ids_list
    lguess = 0
    if guess_tokens is not None:
        ids += list(guess_tokens)
        pos += list(range(lst_id + 1, lst_id + 1 + gs)) * (len(guess_tokens) // gs)
        lguess = len(guess_tokens)
    return StepLayout(ids=ids, positions=pos, n_input=len(input_ids), level_sizes=level_sizes, lguess=lguess,
                      is_prefill=past_tokens[1] is None, window=len(past_tokens[fill_level]))



def flash_row_order(n_input: int, level_sizes: Sequence[int], lguess: int) -> List[int]:
    """Row order the reference feeds its flash kernel (lade/models/modeling_llama.py:1471-1485, `swap_axis_for_flash`):
    [inputs | L0 | levels >= 1 interleaved column-major | candidates].  Returns perm with perm[r_flash] = r_eager
    (the index of the same token in the eager order [inputs | L0 | L1 | ... | candidates])."""
    perm = list(range(n_input + level_sizes[0]))
    base = n_input + level_sizes[0]
    nl = len(level_sizes) - 1
    if nl > 0:
        s = level_sizes[1]
        assert all(x == s for x in level_sizes[1:]), "the flash order needs equal level lengths (np.array(...).transpose())"
        for w in range(s):
            for l in range(nl):
                perm.append(base + l * s + w)
    T = n_input + sum(level_sizes) + lguess
    perm += list(range(T - lguess, T))
    return perm


def mask_visible(q: int, c: int, T: int, s: int, lguess: int, gs: int, level_offset: int, dist_offset: int) -> bool:
    """Closed form of `j_make_causal_mask_multilevel` on the T x T new-token block
    (lade/models/modeling_llama.py:115-207); every row also sees all P cache columns (:204-205).

    rows < A = level_offset+dist_offset : causal (:189-192)
    level rows (block ll, column i)     : every col < A (:195), block-0 causal prefix j<=i (:201),
                                          own column i of blocks 1..ll (:202-203)
    candidate rows (cand, pos)          : cols <= level_offset (:184), own candidate causal (:142-181)
    """
    A = level_offset + dist_offset
    if q >= T - lguess:                       # verification branch
        k = q - (T - lguess)
        cand, pos = divmod(k, gs)
        if c <= level_offset:
            return True
        base = T - lguess + cand * gs
        return base <= c <= base + pos
    if q < A:
        return c <= q
    ll, i = divmod(q - A, s)
    if c < A:
        return True
    if c >= T - lguess:
        return False
    r, j = divmod(c - A, s)
    if r == 0:
        return j <= i
    return j == i and r <= ll


def dense_mask(layout: StepLayout, P: int, gs: int) -> np.ndarray:
    """bool [T, P+T], True = visible.  Prefill = plain causal (modeling_llama.py:124-130)."""
    T = layout.T
    m = np.zeros((T, P + T), dtype=bool)
    m[:, :P] = True
    if layout.is_prefill:
        m[:, P:] = np.tril(np.ones((T, T), dtype=bool))
        return m
    s = layout.level_sizes[-1]
    lo, do = layout.level_offset, layout.dist_offset
    for q in range(T):
        for c in range(T):
            m[q, P + c] = mask_visible(q, c, T, s, layout.lguess, gs, lo, do)
    return m

# --------------------------------------------------------------------------------------
# 6. Llama-shaped model step in fp32  (lade/models/modeling_llama.py:222-227, 233-266,
#    321-346, 360-380, 461-563, 1108-1254)
# --------------------------------------------------------------------------------------

def rope_tables(d: int, max_pos: int, theta: float = 10000.0):
    """cos/sin tables [max_pos, d], fp32 (modeling_llama.py:238-256)."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, d, 2).float() / d))
    t = torch.arange(max_pos, dtype=inv_freq.dtype)
    freqs = torch.outer(t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def apply_rope(x, cos, sin, positions):
    """x [h, T, d]; gathered, possibly non-monotone positions (modeling_llama.py:321-346)."""
    c = cos[positions].unsqueeze(0)
    s = sin[positions].unsqueeze(0)
    return (x * c) + (rotate_half(x) * s)


def attention_dense(q, k, v, visible, scale=None):
    """O = softmax_fp32(Q K^T / sqrt(d) + mask) V, mask = 0 / finfo.min (modeling_llama.py:520-541).
    q [H,T,d]; k,v [Hkv,S,d]; visible bool [T,S].  GQA via repeat (:383-392)."""
    H, T, d = q.shape
    Hkv = k.shape[0]
    n_rep = H // Hkv
    if n_rep > 1:
        k = k[:, None].expand(Hkv, n_rep, *k.shape[1:]).reshape(H, *k.shape[1:])
        v = v[:, None].expand(Hkv, n_rep, *v.shape[1:]).reshape(H, *v.shape[1:])
    scores = torch.matmul(q, k.transpose(1, 2)) / math.sqrt(d) if scale is None else torch.matmul(q, k.transpose(1, 2)) * scale
    add = torch.zeros(visible.shape, dtype=torch.float32)
    add[~torch.as_tensor(visible)] = torch.finfo(torch.float32).min
    scores = scores.float() + add[None]
    p = torch.softmax(scores, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(p, v)


class OracleLlama:
    """Llama-shaped decoder, fp32 CPU.  `weights`: dict of tensors
    embed [V,hid], norm [hid], lm_head [V,hid], layers.{i}.{ln1,wq,wk,wv,wo,ln2,wg,wu,wd}
    (nn.Linear layout [out,in]); `cfg`: hidden, inter, layers, heads, kv_heads, head_dim, vocab,
    eps, rope_theta, max_pos."""

    def __init__(self, cfg: dict, weights: Dict[str, torch.Tensor], dtype=torch.float32):
        self.cfg = dict(cfg)
        self.dtype = dtype
        self.w = {k: torch.as_tensor(v).to(dtype) for k, v in weights.items()}
        self.L = cfg["layers"]
        self.H, self.Hkv, self.d = cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
        cos, sin = rope_tables(self.d, cfg["max_pos"], cfg.get("rope_theta", 10000.0))
        self.cos, self.sin = cos.to(dtype), sin.to(dtype)
        # LlamaDynamicNTKScalingRotaryEmbedding (modeling_llama.py:292-318): the tables are REBUILT, with a base that grows with the
        # sequence, whenever a step's kv_seq_len = P + T (:502-510) exceeds the longest length seen so far; cached K rows keep the
        # rotation they were written with.  max_seq_len_cached starts at max_position_embeddings (:243-246).
        sc = cfg.get("rope_scaling") or {}
        self._ntk_factor = float(sc["factor"]) if sc.get("type", sc.get("rope_type")) == "dynamic" else None
        self._rope_cached_len = cfg["max_pos"]
        self.vocab_size = cfg["vocab"]

    def new_cache(self):
        return [[torch.zeros(self.Hkv, 0, self.d, dtype=self.dtype), torch.zeros(self.Hkv, 0, self.d, dtype=self.dtype)]
                for _ in range(self.L)]

    def _rope_update(self, seq_len: int) -> None:
        """_set_cos_sin_cache of the dynamic-NTK embedding (modeling_llama.py:299-316), called like forward() calls it (:260-261)"""
        if seq_len <= self._rope_cached_len:
            return
        self._rope_cached_len = seq_len
        d, mp = self.d, self.cfg["max_pos"]
        base = self.cfg.get("rope_theta", 10000.0)
        if seq_len > mp:
            base = base * ((self._ntk_factor * seq_len / mp) - (self._ntk_factor - 1)) ** (d / (d - 2))
        cos, sin = rope_tables(d, seq_len, base)
        self.cos, self.sin = cos.to(self.dtype), sin.to(self.dtype)

    def _rms(self, x, w):
        v = x.float().pow(2).mean(-1, keepdim=True)
        return w * (x.float() * torch.rsqrt(v + self.cfg["eps"])).to(x.dtype)

    def forward(self, ids: Sequence[int], positions: Sequence[int], visible: np.ndarray, cache) -> torch.Tensor:
        """Appends the T new K/V rows to `cache` (torch.cat semantics, modeling_llama.py:513-516) and
        returns the final-normed hidden states [T, hid]."""
        w = self.w
        T = len(ids)
        if self._ntk_factor is not None:
            self._rope_update(cache[0][0].shape[1] + T)
        pos = torch.as_tensor(list(positions), dtype=torch.long)
        x = w["embed"][torch.as_tensor(list(ids), dtype=torch.long)]
        for i in range(self.L):
            p = f"layers.{i}."
            h = self._rms(x, w[p + "ln1"])
            q = (h @ w[p + "wq"].T).view(T, self.H, self.d).transpose(0, 1)
            k = (h @ w[p + "wk"].T).view(T, self.Hkv, self.d).transpose(0, 1)
            v = (h @ w[p + "wv"].T).view(T, self.Hkv, self.d).transpose(0, 1)
            q = apply_rope(q, self.cos, self.sin, pos)
            k = apply_rope(k, self.cos, self.sin, pos)
            cache[i][0] = torch.cat([cache[i][0], k], dim=1)
            cache[i][1] = torch.cat([cache[i][1], v], dim=1)
            o = attention_dense(q, cache[i][0], cache[i][1], visible)
            o = o.transpose(0, 1).reshape(T, self.H * self.d)
            x = x + o @ w[p + "wo"].T
            h = self._rms(x, w[p + "ln2"])
            a = torch.nn.functional.silu(h @ w[p + "wg"].T) * (h @ w[p + "wu"].T)
            x = x + a @ w[p + "wd"].T
        return self._rms(x, w["norm"])

    def logits(self, hidden_rows: torch.Tensor) -> torch.Tensor:
        return (hidden_rows @ self.w["lm_head"].T).float()

# --------------------------------------------------------------------------------------
# 7. one model step = jforward_multilevel  (lade/models/modeling_llama.py:1381-1608)
# --------------------------------------------------------------------------------------

@dataclass
class StepOut:
    out_logits: torch.Tensor          # [V]     row n_input-1                 (:1578)
    inp_logits: torch.Tensor          # [window, V]  last level's rows        (:1581-1606)
    guess_logits: Optional[torch.Tensor]  # [g*gs, V]                          (:1592)
    kvcache_len: int                  # P + n_input                           (:1570)
    step_len: int                     # P + T                                 (:1513, :1571)
    layout: StepLayout = None


def model_step(model: OracleLlama, cache, input_ids, position_ids, past_tokens, guess_tokens, fill_level, gs) -> StepOut:
    P = cache[0][0].shape[1]
    lay = build_step_layout(input_ids, position_ids, past_tokens, guess_tokens, fill_level, gs)
    vis = dense_mask(lay, P, gs)
    hid = model.forward(lay.ids, lay.positions, vis, cache)
    T, lg = lay.T, lay.lguess
    out_logits = model.logits(hid[lay.n_input - 1])
    inp_logits = model.logits(hid[T - lg - lay.window: T - lg])
    guess_logits = model.logits(hid[T - lg:]) if lg > 0 else None
    return StepOut(out_logits, inp_logits, guess_logits, P + lay.n_input, P + T, lay)


def kv_commit(cache, kvcache_len: int, step_len: int, lguess: int, max_hit: int, max_hit_idx: int, gs: int):
    """lade/decoding.py:1154-1163: copy the accepted candidate's first max_hit K/V rows into place
    and truncate to kvcache_len+max_hit."""
    off = step_len - lguess + max_hit_idx * gs if max_hit > 0 else 0
    for kv in cache:
        for t in (0, 1):
            if max_hit > 0:
                kv[t][:, kvcache_len:kvcache_len + max_hit] = kv[t][:, off:off + max_hit].clone()
            kv[t] = kv[t][:, :kvcache_len + max_hit]


def kv_truncate(cache, n: int):
    for kv in cache:
        kv[0] = kv[0][:, :n]
        kv[1] = kv[1][:, :n]

# --------------------------------------------------------------------------------------
# 8. greedy lookahead loop, single rank or R simulated lookahead-parallel ranks
#    (lade/decoding.py:697-1259)
# --------------------------------------------------------------------------------------

@dataclass
class StepTrace:
    ids: List[int]
    positions: List[int]
    n_input: int
    level_sizes: List[int]
    lguess: int
    first_guess: int
    max_hit: int
    max_hit_idx: int
    hits: List[int]
    accepted: List[int]
    kv_len_after: int
    past_tokens_after: list
    rank_ids: list = None             # R>1: [ids of rank r]
    rank_positions: list = None
    kvcache_len: int = 0
    step_len: int = 0
    P: int = 0


@dataclass
class GenResult:
    tokens: List[int]                 # prompt + generated (trimmed like the reference)
    steps: int
    generated: int
    trace: List[StepTrace] = field(default_factory=list)
    token_map: dict = None


def _argmax_rows(logits: torch.Tensor) -> List[int]:
    return torch.argmax(logits, dim=-1).tolist()


def lookahead_greedy(model: OracleLlama, prompt: Sequence[int], W: int, N: int, G: int, max_length: int,
                     rng: _random.Random, eos_token_id: Optional[int] = None, pool_from_prompt: bool = False,
                     R: int = 1, keep_trace: bool = True) -> GenResult:
    """`jacobi_greedy_search_multilevel` (lade/decoding.py:697-1259).  `max_length` = total length
    at which MaxLengthCriteria fires (prompt + new tokens).  `rng` replaces the global `random`
    (window init consumes W+N-3 `choice` draws, :887-902).  R>1 simulates the R lookahead-parallel
    ranks in one process: every rank keeps its own KV cache; the object collectives of
    :906,1024,1045,1057,1090,1096,1106 become plain list exchanges."""
    gs = N - 1
    all_old_tokens = list(prompt)
    init_len = len(all_old_tokens)
    input_ids = list(prompt)
    attn_len = len(prompt)                      # model_kwargs["attention_mask"].size(1)
    past_tokens = [[rng.choice(all_old_tokens) for _ in range(W + N - 3)]] + [None] * (N - 2)
    fill_level, steps, guess_skip_dist = 0, 0, 0
    token_map: dict = {}
    lst_token = None
    caches = [model.new_cache() for _ in range(R)]
    have_cache = False
    if pool_from_prompt:
        fill_pool_with_prompt(all_old_tokens, token_map, N, G)
    trace: List[StepTrace] = []

    while True:
        # inputs (decoding.py:935-943, modeling_llama.py:1610-1638): positions = cumsum(mask)-1
        if not have_cache:
            in_ids = list(input_ids)
            in_pos = list(range(attn_len))
        else:
            in_ids = input_ids[-1 - guess_skip_dist:]
            in_pos = list(range(attn_len))[-1 - guess_skip_dist:]
        window_full = past_tokens[N - 2] is not None
        guess_all = pool_lookup(token_map, lst_token, window_full, G)

        outs: List[StepOut] = []
        guess_r: List[Optional[List[int]]] = []
        for r in range(R):
            if R > 1:
                g_r = lp_guess_shard(guess_all, gs, R, r)
                pt_r, _, _ = lp_window_shard(past_tokens, R, r)
            else:
                g_r, pt_r = guess_all, past_tokens
            guess_r.append(g_r)
            outs.append(model_step(model, caches[r], in_ids, in_pos, pt_r, g_r, fill_level, gs))
        steps += 1
        have_cache = True

        first_guess = int(torch.argmax(outs[0].out_logits).item())   # rank 0 broadcasts (:1024)
        max_hit, max_hit_idx = 0, 0
        hits = [first_guess] + [0] * (gs - 1)

        if past_tokens[1] is None:                                   # :1038-1048
            assert fill_level == 0
            window_fill_first(past_tokens, _argmax_rows(outs[R - 1].inp_logits))   # last rank broadcasts (:1045)
            fill_level += 1
        elif past_tokens[N - 2] is None:                             # :1049-1066
            cur: List[int] = []
            for r in range(R):
                cur += _argmax_rows(outs[r].inp_logits)
            window_fill(past_tokens, fill_level, cur)
            fill_level += 1
        else:                                                        # :1067-1130
            per_rank = []
            for r in range(R):
                if guess_r[r] is not None:
                    per_rank.append(greedy_verify(first_guess, guess_r[r], _argmax_rows(outs[r].guess_logits), gs))
                else:
                    per_rank.append((0, 0, list(hits)))
            mh_all = [p[0] for p in per_rank]
            max_hit = max(mh_all)
            win = mh_all.index(max_hit)                              # lowest rank with the max (:1092)
            max_hit_idx = per_rank[win][1]
            if max_hit > 0:
                hits = list(per_rank[win][2])
            new_results: List[int] = []
            for r in range(R):
                new_results += _argmax_rows(outs[r].inp_logits)
            assert len(past_tokens[N - 2]) == W and len(new_results) == W
            update_token_map(token_map, lst_token, past_tokens, new_results, N, W, G)
            window_roll(past_tokens, new_results, N)

        if max_hit > 0:
            attn_len += max_hit                                      # :1134-1140
        kvcache_len, step_len = outs[0].kvcache_len, None
        if R > 1 and max_hit > 0:                                    # :1148-1153
            guess_skip_dist = max_hit
            for r in range(R):
                kv_truncate(caches[r], outs[r].kvcache_len)
        else:
            guess_skip_dist = 0
            for r in range(R):
                lg_r = outs[r].layout.lguess
                kv_commit(caches[r], outs[r].kvcache_len, outs[r].step_len, lg_r, max_hit, max_hit_idx, gs)

        lst_token = hits[max_hit]
        finished = False
        for hit_idx in range(max_hit + 1):                           # :1167-1177
            if eos_token_id is not None and hits[hit_idx] == eos_token_id:
                all_old_tokens.append(hits[hit_idx])
                max_hit = hit_idx
                finished = True
                break
            else:
                all_old_tokens.append(hits[max_hit])                 # reference quirk (Appendix B.2)
                if pool_from_prompt:
                    append_new_generated_pool(all_old_tokens[-N:], token_map, N, G)
        accepted = hits[:max_hit + 1]
        input_ids = input_ids + accepted
        attn_len += 1                                                # _update_model_kwargs_for_generation
        if keep_trace:
            lay = outs[0].layout
            trace.append(StepTrace(ids=list(lay.ids), positions=list(lay.positions), n_input=lay.n_input,
                                   level_sizes=list(lay.level_sizes), lguess=lay.lguess, first_guess=first_guess,
                                   max_hit=max_hit, max_hit_idx=max_hit_idx, hits=list(hits), accepted=list(accepted),
                                   kv_len_after=caches[0][0][0].shape[1],
                                   past_tokens_after=[None if p is None else list(p) for p in past_tokens],
                                   rank_ids=[list(o.layout.ids) for o in outs],
                                   rank_positions=[list(o.layout.positions) for o in outs],
                                   kvcache_len=outs[0].kvcache_len, step_len=outs[0].step_len,
                                   P=outs[0].kvcache_len - lay.n_input))
        if len(input_ids) >= max_length:
            finished = True
        if finished:
            break

    input_ids = input_ids[:max_length]                               # :1221-1229
    all_old_tokens = all_old_tokens[:max_length]
    return GenResult(tokens=input_ids, steps=steps, generated=len(all_old_tokens) - init_len, trace=trace,
                     token_map=token_map)


def plain_greedy(model: OracleLlama, prompt: Sequence[int], max_length: int, eos_token_id: Optional[int] = None) -> List[int]:
    """Ordinary one-token-per-step greedy decoding with a KV cache -- the sequence lookahead must
    reproduce exactly (README.md:132, minimal.py:55)."""
    cache = model.new_cache()
    ids = list(prompt)
    feed, pos0 = list(prompt), 0
    while len(ids) < max_length:
        T = len(feed)
        P = cache[0][0].shape[1]
        vis = np.zeros((T, P + T), dtype=bool)
        vis[:, :P] = True
        vis[:, P:] = np.tril(np.ones((T, T), dtype=bool))
        hid = model.forward(feed, list(range(pos0, pos0 + T)), vis, cache)
        nxt = int(torch.argmax(model.logits(hid[-1])).item())
        ids.append(nxt)
        pos0 += T
        feed = [nxt]
        if eos_token_id is not None and nxt == eos_token_id:
            break
    return ids

# --------------------------------------------------------------------------------------
# 9. sampling lookahead loop  (lade/decoding.py:137-692) -- single rank only
# --------------------------------------------------------------------------------------

def warp_logits(logits: torch.Tensor, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0) -> torch.Tensor:
    """HF TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper on [..., V] fp32 logits
    (the only warpers the reference admits, lade/decoding.py:375-377)."""
    x = logits
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


def sample_verify(probs_next: torch.Tensor, guess_probs: torch.Tensor, guess_tokens: Sequence[int], gs: int,
                  rng: _random.Random, multinomial):
    """SpecInfer-style multi-candidate rejection sampling (lade/decoding.py:484-540).
    probs_next [V] (mutated copy), guess_probs [g*gs, V].  `rng.random()` is drawn once per trial,
    `multinomial(p)` once on the first position where every surviving candidate is rejected.
    Returns (hits, max_hit_idx)."""
    probs_next = probs_next.clone()
    hits: List[int] = []
    guess_indices = list(range(guess_probs.shape[0] // gs))
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
            probs_next = guess_probs[guess_offset + idx_in_ngram].clone()
            continue
        hits.append(int(multinomial(probs_next)))
        break
    return hits, max_hit_idx


def lookahead_sample(model: OracleLlama, prompt: Sequence[int], W: int, N: int, G: int, max_length: int,
                     rng: _random.Random, torch_gen: torch.Generator, temperature: float = 1.0, top_k: int = 0,
                     top_p: float = 1.0, eos_token_id: Optional[int] = None, pool_from_prompt: bool = False,
                     keep_trace: bool = True) -> GenResult:
    """`jacobi_sample_multilevel` (lade/decoding.py:137-692)."""
    gs = N - 1
    all_old_tokens = list(prompt)
    init_len = len(all_old_tokens)
    input_ids = list(prompt)
    attn_len = len(prompt)
    set_token = lambda: rng.choice(all_old_tokens)
    past_tokens = [[set_token() for _ in range(W + N - 3)]] + [None] * (N - 2)
    fill_level, steps = 0, 0
    token_map: dict = {}
    lst_token = None
    cache = model.new_cache()
    have_cache = False
    if pool_from_prompt:
        fill_pool_with_prompt(all_old_tokens, token_map, N, G)
    multinomial = lambda p: torch.multinomial(p, num_samples=1, generator=torch_gen).item()
    warp = lambda x: warp_logits(x, temperature, top_k, top_p)
    trace: List[StepTrace] = []

    while True:
        if not have_cache:
            in_ids, in_pos = list(input_ids), list(range(attn_len))
        else:
            in_ids, in_pos = input_ids[-1:], [attn_len - 1]
        guess_tokens = pool_lookup(token_map, lst_token, past_tokens[N - 2] is not None, G)
        out = model_step(model, cache, in_ids, in_pos, past_tokens, guess_tokens, fill_level, gs)
        steps += 1
        have_cache = True
        next_token_scores = warp(out.out_logits)
        max_hit, max_hit_idx = 0, 0
        if past_tokens[1] is None:                                   # :453-463
            probs = torch.softmax(next_token_scores, dim=-1)
            hits = [int(multinomial(probs))]
            window_fill_first(past_tokens, _argmax_rows(out.inp_logits))
            fill_level += 1
        elif past_tokens[N - 2] is None:                             # :464-476
            probs = torch.softmax(next_token_scores, dim=-1)
            hits = [int(multinomial(probs))]
            window_fill(past_tokens, fill_level, _argmax_rows(out.inp_logits))
            fill_level += 1
        else:
            if guess_tokens is not None:                             # :484-540
                probs_next = torch.softmax(next_token_scores, dim=-1)
                guess_probs = torch.softmax(warp(out.guess_logits), dim=-1)
                hits, max_hit_idx = sample_verify(probs_next, guess_probs, guess_tokens, gs, rng, multinomial)
                max_hit = len(hits) - 1
            else:
                probs_next = torch.softmax(next_token_scores, dim=-1)
                hits = [int(multinomial(probs_next))]
            new_results = _argmax_rows(out.inp_logits)
            update_token_map(token_map, lst_token, past_tokens, new_results, N, W, G)
            window_roll(past_tokens, new_results, N)
            if max_hit > 0:
                attn_len += max_hit
            if eos_token_id is not None:                             # :578-580
                filter_window(past_tokens[N - 2], eos_token_id, set_token)
        kv_commit(cache, out.kvcache_len, out.step_len, out.layout.lguess, max_hit, max_hit_idx, gs)   # :582-590
        lst_token = hits[max_hit]
        finished = False
        for hit_idx in range(max_hit + 1):                           # :594-604
            if eos_token_id is not None and hits[hit_idx] == eos_token_id:
                all_old_tokens.append(hits[hit_idx])
                max_hit = hit_idx
                finished = True
                break
            else:
                all_old_tokens.append(hits[hit_idx])
                if pool_from_prompt:
                    append_new_generated_pool(all_old_tokens[-N:], token_map, N, G)
        accepted = hits[:max_hit + 1]
        input_ids = input_ids + accepted
        attn_len += 1
        if keep_trace:
            lay = out.layout
            trace.append(StepTrace(ids=list(lay.ids), positions=list(lay.positions), n_input=lay.n_input,
                                   level_sizes=list(lay.level_sizes), lguess=lay.lguess, first_guess=hits[0],
                                   max_hit=max_hit, max_hit_idx=max_hit_idx, hits=list(hits), accepted=list(accepted),
                                   kv_len_after=cache[0][0].shape[1],
                                   past_tokens_after=[None if p is None else list(p) for p in past_tokens],
                                   kvcache_len=out.kvcache_len, step_len=out.step_len, P=out.kvcache_len - lay.n_input))
        if len(input_ids) >= max_length:
            finished = True
        if finished:
            break
    input_ids = input_ids[:max_length]
    all_old_tokens = all_old_tokens[:max_length]
    return GenResult(tokens=input_ids, steps=steps, generated=len(all_old_tokens) - init_len, trace=trace,
                     token_map=token_map)
