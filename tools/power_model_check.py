"""A check of DESIGN 4.6's power model (P = 291 W idle + 150 W per TB/s streamed + 0.67 W per TFLOP/s multiplied, fitted on the weight-streaming GEMM) in the regime it
was NOT fitted on: the library's compute-bound GEMM (hipBLASLt through torch.matmul, 7B qkv shape at 512 / 2304 rows, rotating weights), sampled with amd-smi while it
replays for seconds.  python tools/power_model_check.py"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from clock_probe_lib import sample

N, K = 12288, 4096
ws = [(torch.randn(N, K, device="cuda") * 0.02).bfloat16() for _ in range(6)]
print("idle:", sample(), flush=True)
for M in (512, 2304, 8192):
    a = torch.randn(M, K, device="cuda").bfloat16()
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    i = [0]

    def run():
        i[0] = (i[0] + 1) % len(ws)
        torch.matmul(a, ws[i[0]].t(), out=out)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(24):
            run()
    samples, stop = [], threading.Event()

    def sampler():
        time.sleep(1.0)
        while not stop.is_set():
            samples.append(sample())
            time.sleep(0.3)
    th = threading.Thread(target=sampler)
    th.start()
    t0, n = time.time(), 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < 4.0:
        for _ in range(10):
            g.replay()
        n += 10
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    us = e0.elapsed_time(e1) * 1e3 / (n * 24)
    tf = 2.0 * M * N * K / us / 1e6
    tbs = (N * K * 2 + M * K * 2 + M * N * 2) / us / 1e6          # weights + activations + output once (a lower bound of the traffic)
    good = [s for s in samples if s and s[0] not in (None, "ERR")]
    pw = sum(s[0] for s in good) / len(good) if good else float("nan")
    ck = [s[1] for s in samples if s and isinstance(s[1], float)]
    print(f"library GEMM M={M:5d} N={N} K={K}: {us:8.1f} us  {tf:6.0f} TFLOP/s  >= {tbs:4.2f} TB/s | power {pw:6.1f} W  gfx clock mean {sum(ck) / len(ck) if ck else float('nan'):7.1f} MHz | "
          f"model 291 + 150 x {tbs:.2f} + 0.67 x {tf:.0f} = {291 + 150 * tbs + 0.67 * tf:6.0f} W", flush=True)
