"""Socket power and GFX clock while WHOLE decode steps of T rows replay (hipGraph of the full model's forward + argmax, causal rows over a 2048-key
cache): is the step itself - not only an isolated GEMM launch (tools/clock_probe.py) - power limited above 96 rows?
    python tools/step_power_probe.py 13b [rows ...]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from clock_probe_lib import sample  # noqa: E402  (tools/ is on sys.path when run as a script)
from lookaheaddecoding_amd import ops
from lookaheaddecoding_amd.engine import StepEngine
from lookaheaddecoding_amd.weights import make_config, random_weights_torch

MODEL = {"7b": "llama2-7b", "13b": "codellama-13b"}[sys.argv[1] if len(sys.argv) > 1 else "13b"]
ROWS = [int(x) for x in sys.argv[2:]] or [1, 60, 92, 120, 150, 180]
SECONDS = float(os.environ.get("SECONDS_PER_CASE", "4"))
P = 2048
cfg = make_config(MODEL)
dev = torch.device("cuda", 0)
w = random_weights_torch(cfg, seed=0, dtype=torch.bfloat16, device=dev)
eng = StepEngine(cfg, w, dtype=torch.bfloat16, device=dev, max_seq=P + 512, max_T=256, consume_weights=True)
del w
print(f"{MODEL} bf16, P={P}: whole steps replayed for {SECONDS:.0f} s each; idle: {sample()}", flush=True)
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
    while time.time() - t0 < SECONDS:
        for _ in range(20):
            g.replay()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    ms = e0.elapsed_time(e1) / n
    good = [s for s in samples if s and s[0] not in (None, "ERR")]
    pw = sum(s[0] for s in good) / len(good) if good else float("nan")
    ck = [s[1] for s in samples if s and isinstance(s[1], float)]
    print(f"  T={T:4d}  {ms:7.3f} ms per step  power {pw:6.1f} W  gfx clock mean {sum(ck) / len(ck) if ck else float('nan'):7.1f} MHz  ({len(samples)} samples)", flush=True)
