"""step(T rows) / step(1 row): the steady forward of a T-row step (causal rows over a 2048-key cache, lm_head + argmax on the last min(T, 31)
rows) as a hipGraph on the full model, for a list of row counts - the cost that decides whether a lookahead step pays.
    python tools/rows_curve.py 7b [rows ...]          LADE_DEBUG=row_classes=r5: the row classes of rounds 1-5 (no 160-row class)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import ops
from lookaheaddecoding_amd.engine import StepEngine
from lookaheaddecoding_amd.weights import make_config, random_weights_torch

MODEL = {"7b": "llama2-7b", "13b": "codellama-13b"}[sys.argv[1] if len(sys.argv) > 1 else "7b"]
ROWS = [int(x) for x in sys.argv[2:]] or [1, 32, 60, 64, 76, 92, 96, 104, 120, 128, 132, 136, 144, 150, 156, 160, 168, 180, 192, 210, 240, 256]
P = 2048
cfg = make_config(MODEL)
dev = torch.device("cuda", 0)
w = random_weights_torch(cfg, seed=0, dtype=torch.bfloat16, device=dev)
eng = StepEngine(cfg, w, dtype=torch.bfloat16, device=dev, max_seq=P + 512, max_T=256, consume_weights=True)
del w
print(f"{MODEL} bf16, P={P}, row classes {eng.ROW_CLASSES}: ms per step (hipGraph, best of 6 x 4 replays)", flush=True)
t1 = None
for T in ROWS:
    mask = ops.StepMask(T=T, P=P, is_prefill=True)
    ids = torch.randint(3, cfg["vocab"], (T,), device=dev, dtype=torch.int32)
    pos = torch.arange(P, P + T, device=dev, dtype=torch.int32)
    n_sel = min(T, 31)
    sel = torch.arange(T - n_sel, T, device=dev, dtype=torch.int32)
    out = torch.zeros(64, dtype=torch.int32, device=dev)

    def run():
        eng.forward(ids, pos, mask, sel, n_sel, argmax_out=out)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    best = 1e9
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 4)
    if t1 is None:
        t1 = best
    cls = next((c for c in eng.ROW_CLASSES if T <= c), None)
    cfgs = " ".join(f"{n}={eng.gemm_cfg.get((n, cls))}" for n in eng.LAYER_GEMMS)
    print(f"  T={T:4d}  {best:7.3f} ms  x{best / t1:5.3f} of the first row count  class {cls}  {cfgs}", flush=True)
