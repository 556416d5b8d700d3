"""A/B of kernel-level switches of the skinny GEMM (LADE_DEBUG=gemm_dbg bits, read once per process) on the four 7B projections at the
configurations the engine's autotune picks, partials left for the consumer (no reduce pass), hipGraph of dependent launches with
rotating weights: python tools/gemm_flags.py   (run once per LADE_DEBUG=gemm_dbg=<bits> value)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import cabi
from lookaheaddecoding_amd.cabi import call, dtype_code, ptr

M = int(os.environ.get("M", "60"))
CFG = {"qkv": (12288, 4096, 2, 2, 192, 0, 4), "o": (4096, 4096, 2, 2, 128, 0, 8), "gate_up": (22016, 4096, 2, 2, 192, 0, 4), "down": (4096, 11008, 2, 1, 128, 0, 8)}
if M > 64:      # the 128-row class: four m-blocks, two per wave
    CFG = {"qkv": (12288, 4096, 4, 2, 192, 0, 4), "o": (4096, 4096, 4, 2, 128, 0, 8), "gate_up": (22016, 4096, 4, 2, 192, 0, 4), "down": (4096, 11008, 4, 2, 128, 0, 8)}


def timeit(fn, reps=40, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


if os.environ.get("SHAPES") == "13b":      # CodeLlama-13B projections at the configurations the autotune picks for 120-row steps
    CFG = {"qkv": (15360, 5120, 4, 4, 256, 0, 4), "o": (5120, 5120, 4, 2, 128, 0, 6), "gate_up": (27648, 5120, 4, 4, 256, 0, 2), "down": (5120, 13824, 4, 1, 192, 0, 9)}
tot = 0.0
line = []
for name, (N, K, mb, mt, bn, nt, S) in CFG.items():
    a = torch.randn(M, K, device="cuda").bfloat16()
    ws = [torch.randn(N, K, device="cuda").bfloat16() * 0.02 for _ in range(max(2, int(600e6 / (N * K * 2))))]
    part = torch.empty(S, M, N, dtype=torch.float32, device="cuda")
    i = [0]

    def mine():
        i[0] = (i[0] + 1) % len(ws)
        call("lade_gemm_skinny", ptr(a), a.stride(0), ptr(ws[i[0]]), ws[i[0]].stride(0), None, 0, ptr(part), M, N, K, S, bn, mb, mt, nt, 0, 0, dtype_code(a))

    t = timeit(mine)
    tot += t
    line.append(f"{name} {t:6.2f} us {N * K * 2 / t / 1e6:5.2f} TB/s")
print(f"LADE_DEBUG=gemm_dbg={cabi.debug('gemm_dbg', '0'):>2s} M={M}: " + " | ".join(line) + f" | sum {tot:6.2f} us", flush=True)
