import os, sys, random
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import torch
from lookaheaddecoding_amd.decoding import LookaheadDecoder
from lookaheaddecoding_amd.engine import StepEngine
from lookaheaddecoding_amd.weights import make_config, random_weights_torch
cfg = make_config("llama2-7b"); cfg["layers"] = 2
w = random_weights_torch(cfg, seed=0, dtype=torch.bfloat16, device="cuda")
eng = StepEngine(cfg, w, dtype=torch.bfloat16, device="cuda", max_seq=1024, max_T=512)
C = 256
eng.zero_projections(("wo", "wd"))
head = eng.embed.clone(); head[:C] = eng.embed[(torch.arange(C, device="cuda") - 1) % C]
eng.lm_head = head
prompt = [i % C for i in range(300)]
n_new = 96
want = [(prompt[-1] + 1 + i) % C for i in range(n_new)]
print("plain ok", eng.plain_greedy(prompt, len(prompt) + 24)[len(prompt):] == want[:24])
for use_graph in (False, True):
    dec = LookaheadDecoder(eng, 15, 5, 15, pool_from_prompt=True, use_graph=use_graph)
    out = dec.greedy(prompt, len(prompt) + n_new, rng=random.Random(1), keep_trace=True)
    got = out.tokens[len(prompt):]
    bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
    print("graph", use_graph, "ok", got == want, "steps", out.steps, "first bad", bad[:5], got[:8], want[:8])
    for t in out.trace[:8]: print("   ", {k: t[k] for k in ("T", "P_before", "max_hit", "accepted", "phase")})
    print("  cfg", {k: v for k, v in eng.gemm_cfg.items()}, "nan in x", torch.isnan(eng.ws_x.float()).any().item())
