"""lade_warp_rows on a Llama-3-class vocabulary (V = 128256: the row's keys live in the output row, csrc/sampling.hip) against the torch
ops it replaces (`sampling.Warper.__call__`: division, topk, sort, softmax, cumsum, scatter, masked_fill), 31 rows (1 + 15 + 15 of a
config-3 step), microseconds per call; and the register form at V = 32000 for scale.   python tools/warp_big_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import ops
from lookaheaddecoding_amd.sampling import Warper


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for V in (32000, 128256):
    lg = (torch.randn(31, V, device="cuda") * 3.0).bfloat16()
    for (temp, k, p) in ((0.8, 50, 0.9), (1.0, 0, 0.9), (0.7, 40, 1.0)):
        w = Warper(temp, k, p)
        out = torch.empty(31, V, dtype=torch.float32, device="cuda")
        t_hip = timed(lambda: ops.warp_rows(lg, 31, 0, temp, k, p, out=out))
        t_torch = timed(lambda: w(lg.float()))
        print(f"V={V:6d} T={temp} top_k={k:3d} top_p={p}: lade_warp_rows {t_hip:8.1f} us   torch warpers {t_torch:8.1f} us", flush=True)
