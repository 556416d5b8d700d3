"""diagnostic (round 5): whole decoding runs with the three forms of RoPE + KV append (0 two launches, 1 every split rebuilds q, 2 producer
work-groups) - where do the K / V rows differ, and is a form deterministic from run to run?"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LADE_TUNE_STEP"] = "0"
import torch
from lookaheaddecoding_amd.decoding import LookaheadDecoder
from lookaheaddecoding_amd.engine import StepEngine
from lookaheaddecoding_amd.weights import make_config, random_weights_torch

cfg = make_config("llama2-7b", layers=2)
w = random_weights_torch(cfg, seed=2, dtype=torch.bfloat16, device="cuda", std=0.03)
rng = random.Random(5)
prompt = [rng.randrange(3, cfg["vocab"]) for _ in range(150)]
Hkv, d, S = cfg["kv_heads"], cfg["head_dim"], 512
outs = {}
for fuse in (0, 1, 1, 2, 0):
    eng = StepEngine(cfg, w, dtype=torch.bfloat16, device="cuda", max_seq=512, max_T=128)
    eng.attn_default = (fuse, 128, 0)
    dec = LookaheadDecoder(eng, 15, 5, 15, use_graph=False)
    o = dec.greedy(prompt, len(prompt) + 24, rng=random.Random(1), keep_trace=True)
    kv = eng.kv.view(eng.L, 2, -1).clone()
    key = (fuse, sum(1 for k in outs if k[0] == fuse))
    outs[key] = (o.tokens, o.steps, kv, [(t["T"], t["P_before"], t["max_hit"]) for t in o.trace])
    print(key, "tokens", len(o.tokens), "steps", o.steps, "trace", outs[key][3][:6], flush=True)
ref = outs[(0, 0)]
n_keep = len(ref[0]) - 1
for key, (tok, steps, kv, tr) in outs.items():
    K, Kr = kv[:, 0].view(-1, Hkv, S, d), ref[2][:, 0].view(-1, Hkv, S, d)
    V, Vr = kv[:, 1].view(-1, Hkv, d, S), ref[2][:, 1].view(-1, Hkv, d, S)
    dk = (K.view(torch.int16) != Kr.view(torch.int16))
    dv = (V.view(torch.int16) != Vr.view(torch.int16))
    rows_k = sorted(set(dk.nonzero()[:, 2].tolist()))
    rows_v = sorted(set(dv.nonzero()[:, 3].tolist()))
    print(key, "tokens equal", tok == ref[0], "| K rows differing:", rows_k[:12], "... of", len(rows_k), "| V columns differing:", rows_v[:12], "... of", len(rows_v),
          "| n_keep", n_keep, "| layers K", sorted(set(dk.nonzero()[:, 0].tolist())), "heads K", sorted(set(dk.nonzero()[:, 1].tolist()))[:8])
