"""The attention launch(es) timed IN CONTEXT (run on the GPU box): a Llama-2-7B-shaped engine with a few layers (weights >> the
Infinity Cache, so every kernel of the step starts as cold as in the full model), one steady lookahead step (T = 60 / 120 tokens at a
2 k cache) replayed as a hipGraph, with and without the attention launches; the difference / layers is what the attention costs inside
a step (launch boundaries, cold instruction cache and cold K/V included) - the number bench.py reports as launch_us_in_step, in seconds
instead of minutes.   python tools/attn_in_step.py [--layers 8] [--T 60 120] [--P 2016] [--splits 0 6 8]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import ops
from lookaheaddecoding_amd.engine import StepEngine
from lookaheaddecoding_amd.weights import make_config, random_weights_torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama2-7b")
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--T", type=int, nargs="+", default=[60])
    ap.add_argument("--P", type=int, default=2016)
    ap.add_argument("--splits", type=int, nargs="+", default=[0])
    ap.add_argument("--reps", type=int, default=60)
    a = ap.parse_args()
    cfg = make_config(a.model, layers=a.layers)
    dev = torch.device("cuda", 0)
    w = random_weights_torch(cfg, seed=0, dtype=torch.bfloat16, device=dev)
    eng = StepEngine(cfg, w, dtype=torch.bfloat16, device=dev, max_seq=a.P + 512, max_T=512, consume_weights=True)
    del w
    eng.kv.normal_()                                   # a cache of P synthetic rows: the timing does not depend on the values
    W, N = (20, 7) if "13b" in a.model else (15, 5)
    gs = N - 1
    for T in a.T:
        g = max(0, (T - (N - 1) * W) // gs)
        mask = ops.StepMask.from_levels(1, [W - 1] + [W] * (N - 2), g * gs, gs, a.P)
        T = mask.T
        ids = torch.randint(3, cfg["vocab"], (T,), dtype=torch.int32, device=dev)
        pos = torch.arange(a.P, a.P + T, dtype=torch.int32, device=dev)
        sel = torch.arange(0, 1 + W, dtype=torch.int32, device=dev)
        alg = 2 * (2 * cfg["kv_heads"] * (a.P + T) * cfg["head_dim"] + 2 * cfg["heads"] * T * cfg["head_dim"])

        def timed(ns):
            for _ in range(3):
                eng.forward(ids, pos, mask, sel, sel.numel(), n_splits=ns)
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph):
                eng.forward(ids, pos, mask, sel, sel.numel(), n_splits=ns)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(a.reps):
                    gph.replay()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / a.reps)
            return best * 1e6

        eng.skip_attn = True
        base = timed(1)
        eng.skip_attn = False
        print(f"T={T} P={a.P} layers={a.layers}: step without attention {base:.1f} us", flush=True)
        for ns in a.splits:
            n = ns if ns > 0 else eng.n_splits_for(T, a.P + T)
            us = (timed(n) - base) / a.layers
            print(f"  splits={n:2d}: attention (+ split merge) in step {us:6.2f} us / layer   {alg / us / 1e3:7.1f} GB/s ({alg / us / 1e3 / 80:5.1f} % of 8 TB/s)", flush=True)

if __name__ == "__main__":
    main()
