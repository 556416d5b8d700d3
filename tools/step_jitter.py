"""Per-step wall time of the config-2 decode loop (Llama-2-7B shape, W15 N5 G15, prompt 2048), step by step: bench.py's blocks of 32
steps show one block in three about 1.2 ms / step slower than its neighbours (`spread.ms_per_step_blocks`), i.e. some single event of
about 38 ms.  This tool times every step on the host (the step ends when the sealed record is in pinned memory), prints the steps
that took more than 1.5 x the median with what the process was doing around them (Python gc generations collected, graph captures,
GEMM tune events), and repeats the run with the collector frozen (`gc.freeze(); gc.disable()`).

    python tools/step_jitter.py [steps]"""
import gc
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd.decoding import LookaheadDecoder
from lookaheaddecoding_amd.engine import StepEngine
from lookaheaddecoding_amd.weights import make_config, random_weights_torch

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 400
W, N, G, PROMPT = 15, 5, 15, 2048
dev = torch.device("cuda", 0)
cfg = make_config("llama2-7b")
max_seq = PROMPT + (STEPS * 2 + 64) * N + (N - 1) * (W + G) + 64
cfg["max_pos"] = max(cfg.get("max_pos", 4096), max_seq)
w = random_weights_torch(cfg, seed=0, dtype=torch.bfloat16, device=dev)
eng = StepEngine(cfg, w, dtype=torch.bfloat16, device=dev, max_seq=max_seq, max_T=512, consume_weights=True)
del w
dec = LookaheadDecoder(eng, W, N, G, use_graph=True)
prompt = torch.randint(3, cfg["vocab"], (PROMPT,), generator=torch.Generator().manual_seed(123)).tolist()

events = []
gc.callbacks.append(lambda phase, info: events.append((time.perf_counter(), f"gc {phase} gen{info['generation']} collected {info.get('collected', 0)}")))
_cap = dec._capture_graphs


def _cap_logged(*a, **k):
    t = time.perf_counter()
    r = _cap(*a, **k)
    events.append((t, f"graph capture ({(time.perf_counter() - t) * 1e3:.1f} ms)"))
    return r


dec._capture_graphs = _cap_logged
_ref = eng._refine_in_step


def _ref_logged(*a, **k):
    t = time.perf_counter()
    r = _ref(*a, **k)
    events.append((t, f"in-step refinement {a} ({(time.perf_counter() - t) * 1e3:.1f} ms)"))
    return r


eng._refine_in_step = _ref_logged


def run(label):
    dec.start(prompt)
    for _ in range(N - 1 + 8):
        dec.step()
    torch.cuda.synchronize()
    del events[:]
    ts = [time.perf_counter()]
    for _ in range(STEPS):
        dec.step()
        ts.append(time.perf_counter())
    torch.cuda.synchronize()
    d = [(b - a) * 1e3 for a, b in zip(ts, ts[1:])]
    med = statistics.median(d)
    slow = [(i, x) for i, x in enumerate(d) if x > 1.5 * med]
    print(f"{label}: {STEPS} steps, median {med:.3f} ms, mean {sum(d) / len(d):.3f} ms, p99 {sorted(d)[int(len(d) * 0.99)]:.3f} ms, max {max(d):.3f} ms; "
          f"{len(slow)} steps above 1.5 x the median, together {sum(x - med for _, x in slow):.1f} ms = {100 * sum(x - med for _, x in slow) / sum(d):.2f} % of the run", flush=True)
    for i, x in slow[:20]:
        near = [e for t, e in events if ts[i] - 1e-3 <= t <= ts[i + 1] + 1e-3]
        print(f"   step {i:4d} (cache length {PROMPT + N - 1 + 8 + i}): {x:8.3f} ms   {near}", flush=True)
    # coarse picture of the step time along the run (clock / cache-length drift): means of 8 slices without the slow steps
    sl = max(1, STEPS // 8)
    print("   slice means (ms, slow steps excluded):", [round(statistics.mean([x for x in d[k:k + sl] if x <= 1.5 * med]), 3) for k in range(0, STEPS, sl)], flush=True)


run("default (collector on, graph buckets captured on demand)")
run("second generate() on the same decoder")
gc.collect()
gc.freeze()
gc.disable()
run("collector frozen + disabled")
gc.enable()
