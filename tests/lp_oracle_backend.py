"""TEST-ONLY stand-in for the HIP kernels of one lookahead-parallel rank, built on the CPU oracle.
Lets the product's LP orchestration (lookaheaddecoding_amd.parallel.greedy_lp: sharding, the one
all-gather per step, the loop) run on CPU under gloo.  Never imported by the product."""
import torch
import torch.distributed as dist

import lade_oracle as O
from lookaheaddecoding_amd.parallel import REC_HEAD, rec_words


class OracleLPBackend:
    def __init__(self, model, W, N, G, pool_from_prompt=False):
        self.model, self.W, self.N, self.G, self.gs = model, W, N, G, N - 1
        self.pool_from_prompt = bool(pool_from_prompt)
        self.wcap = W + N - 3
        self.rw = rec_words(self.gs, self.wcap, G)

    def begin(self, prompt, window0, eos):
        self.cache = self.model.new_cache()
        self.tokens = list(prompt)
        self.attn_len = len(prompt)
        self.past = [list(window0)] + [None] * (self.N - 2)
        self.old_tail = list(prompt)[-self.N:]
        self.token_map = {}
        if self.pool_from_prompt:
            O.fill_pool_with_prompt(self.tokens, self.token_map, self.N, self.G)
        self.lst_token = None
        self.guess_all = None
        self.fill_level = 0

    def local_step(self, phase, P, n_input, level_lens, c0, c1, g, glo, ghi):
        gs = self.gs
        assert self.cache[0][0].shape[1] == P
        assert [len(p) for p in self.past if p is not None] == list(level_lens)
        in_ids = self.tokens[-n_input:] if phase != 0 else list(self.tokens)
        in_pos = list(range(self.attn_len))[-n_input:] if phase != 0 else list(range(self.attn_len))
        pt = [self.past[0][: c1 - 1]] + [None if p is None else p[c0:c1] for p in self.past[1:]]
        guess = None
        if phase == 2 and ghi > glo:
            guess = self.guess_all[glo * gs: ghi * gs]
        out = O.model_step(self.model, self.cache, in_ids, in_pos, pt, guess, self.fill_level, gs)
        self.kvcache_len = out.kvcache_len
        fg = int(torch.argmax(out.out_logits).item())
        inp = torch.argmax(out.inp_logits, dim=-1).tolist()
        am_guess = torch.argmax(out.guess_logits, dim=-1).tolist() if guess else []
        rec = torch.zeros(self.rw, dtype=torch.int32)
        rec[0], rec[1], rec[2] = fg, len(inp), len(am_guess) // gs
        rec[REC_HEAD:REC_HEAD + len(inp)] = torch.tensor(inp, dtype=torch.int32)
        if am_guess:
            rec[REC_HEAD + self.wcap:REC_HEAD + self.wcap + len(am_guess)] = torch.tensor(am_guess, dtype=torch.int32)
        self.last_ids = out.layout.ids
        return rec

    def apply(self, all_rec, R, phase):
        gs, N, W, G = self.gs, self.N, self.W, self.G
        recs = all_rec.view(R, self.rw).tolist()
        fg = recs[0][0]
        toks = []
        for r in (range(R - 1, R) if phase == 0 else range(R)):
            n = recs[r][1]
            toks += recs[r][REC_HEAD:REC_HEAD + n]
        # verification on the gathered argmax rows against rank 0's first token (lade/decoding.py:1024, :1071-1096)
        max_hit, win, hits = 0, 0, [fg] + [0] * (gs - 1)
        if phase == 2 and self.guess_all:
            am_all = []
            for r in range(R):
                am_all += recs[r][REC_HEAD + self.wcap:REC_HEAD + self.wcap + recs[r][2] * gs]
            max_hit, idx, hits = O.greedy_verify(fg, self.guess_all, am_all, gs)
            cnt = (len(self.guess_all) // gs + R - 1) // R
            win = idx // cnt if max_hit > 0 else 0
            hits = list(hits) + [0] * (gs - len(hits))
        if phase == 2:
            O.update_token_map(self.token_map, self.lst_token, self.past, toks, N, W, G)
            O.window_roll(self.past, toks, N)
        elif phase == 0:
            O.window_fill_first(self.past, toks)
            self.fill_level += 1
        else:
            O.window_fill(self.past, self.fill_level, toks)
            self.fill_level += 1
        O.kv_truncate(self.cache, self.kvcache_len)
        self.lst_token = hits[max_hit]
        self.tokens += hits[:max_hit + 1]
        self.attn_len += max_hit + 1
        if self.pool_from_prompt:                     # lade/decoding.py:1167-1177: hits[max_hit] once per accepted index
            for _ in range(max_hit + 1):
                self.old_tail = (self.old_tail + [hits[max_hit]])[-N:]
                O.append_new_generated_pool(self.old_tail, self.token_map, N, G)
        self.guess_all = O.pool_lookup(self.token_map, self.lst_token, self.past[N - 2] is not None, G)
        g_next = len(self.guess_all) // gs if self.guess_all else 0
        return [max_hit, max_hit + 1, 0, g_next, self.kvcache_len, win, fg, len(toks)] + list(hits) + [0] * 8

    def new_gather_buffer(self, R):
        return torch.zeros(R * self.rw, dtype=torch.int32)

    def broadcast_window(self, window0, lp):
        t = torch.tensor(window0, dtype=torch.int32)
        dist.broadcast(t, src=0, group=lp.group)
        return t.tolist()
