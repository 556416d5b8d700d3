"""diagnostic (round 5): ONE engine (weights K-tile-major only: no library GEMM), the three forms of RoPE + KV append in turn - K / V of the prompt
rows right after the prefill and after the whole run: which form differs where, and does a decode step ever touch a committed row?"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LADE_TUNE_STEP"] = "0"
os.environ["LADE_W_KTILE"] = "only"
import torch
from lookaheaddecoding_amd.decoding import LookaheadDecoder
from lookaheaddecoding_amd.engine import StepEngine
from lookaheaddecoding_amd.weights import make_config, random_weights_torch

cfg = make_config("llama2-7b", layers=2)
w = random_weights_torch(cfg, seed=2, dtype=torch.bfloat16, device="cuda", std=0.03)
rng = random.Random(5)
prompt = [rng.randrange(3, cfg["vocab"]) for _ in range(150)]
Hkv, d, S = cfg["kv_heads"], cfg["head_dim"], 512
eng = StepEngine(cfg, w, dtype=torch.bfloat16, device="cuda", max_seq=512, max_T=128, consume_weights=True)
LookaheadDecoder(eng, 15, 5, 15).greedy(prompt, len(prompt) + 8, rng=random.Random(1))
POISON = os.environ.get("DIAG_POISON", "")


def poison():
    """what a recycled allocation looks like: every workspace the engine obtained with torch.empty filled with a chosen pattern"""
    if not POISON:
        return
    val = float("nan") if POISON == "nan" else float(POISON)
    for name in ("ws_x", "ws_h", "ws_r", "ws_qkv", "ws_o", "ws_gu", "ws_a", "ws_q", "part_o", "part_ml", "ws_part"):
        t = getattr(eng, name, None)
        if t is not None and (not os.environ.get("DIAG_ONLY") or name in os.environ["DIAG_ONLY"].split(",")):
            t.fill_(val)


def snap():
    kv = eng.kv.view(eng.L, 2, -1)
    return kv[:, 0].view(-1, Hkv, S, d)[:, :, :150].clone(), kv[:, 1].view(-1, Hkv, d, S)[:, :, :, :150].clone()


def diff(a, b):
    dk = (a[0].view(torch.int16) != b[0].view(torch.int16)).nonzero()
    dv = (a[1].contiguous().view(torch.int16) != b[1].contiguous().view(torch.int16)).nonzero()
    return (sorted(set(dk[:, 2].tolist()))[:8], sorted(set(dk[:, 0].tolist())), len(dk)), (sorted(set(dv[:, 3].tolist()))[:8], sorted(set(dv[:, 0].tolist())), len(dv))


ref_pre = None
for fuse in (0, 1, 2):
    for graph in (False, True):
        eng.attn_default = (fuse, 128, 0)
        dec = LookaheadDecoder(eng, 15, 5, 15, use_graph=graph)
        poison()
        dec.start(prompt, rng=random.Random(1))
        dec.step()                                   # prefill
        torch.cuda.synchronize()
        pre = snap()
        if ref_pre is None:
            ref_pre = pre
        n = 1
        while len(dec.tokens) < len(prompt) + 24:
            dec.step()
            n += 1
        torch.cuda.synchronize()
        post = snap()
        print(f"fuse {fuse} graph {graph}: steps {n} | prefill vs first run's prefill: K {diff(pre, ref_pre)[0]} V {diff(pre, ref_pre)[1]} | after the run vs its own prefill: K {diff(post, pre)[0]} V {diff(post, pre)[1]}",
              "| gemm wqkv:128", eng.gemm_cfg.get(("wqkv", 128)), "wqkv:64", eng.gemm_cfg.get(("wqkv", 64)), flush=True)
