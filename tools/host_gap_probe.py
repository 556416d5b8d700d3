"""Where the host's share of a steady step goes (bench.py `step_gpu_only`: 25-45 us per step at config 2): wall-clock stamps around the
pieces of LookaheadDecoder.step() in the hipGraph loop - python before the replay call, the replay call itself (hipGraphLaunch), the
poll, python after the poll.  The GPU idles from the record's arrival to the next graph's first kernel = after + loop + before + the
launch latency of the graph."""
import argparse
import os
import random
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama2-7b")
    ap.add_argument("--steps", type=int, default=96)
    ap.add_argument("--prompt-len", type=int, default=2048)
    a = ap.parse_args()
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    from lookaheaddecoding_amd.weights import make_config, random_weights_torch
    dev = torch.device("cuda", 0)
    cfg = make_config(a.model)
    W, N, G = 15, 5, 15
    max_seq = a.prompt_len + (a.steps + 40) * N + (N - 1) * (W + G) + 64
    cfg["max_pos"] = max(cfg.get("max_pos", 4096), max_seq)
    weights = random_weights_torch(cfg, seed=0, dtype=torch.bfloat16, device=dev)
    eng = StepEngine(cfg, weights, dtype=torch.bfloat16, device=dev, max_seq=max_seq, max_T=2304, consume_weights=True)
    del weights
    dec = LookaheadDecoder(eng, W, N, G, use_graph=True)
    prompt = torch.randint(3, cfg["vocab"], (a.prompt_len,), generator=torch.Generator().manual_seed(123)).tolist()
    dec.start(prompt, rng=random.Random(1))
    for _ in range(N + 8):
        dec.step()
    marks = {}
    ns = time.perf_counter_ns
    orig_replay = torch.cuda.CUDAGraph.replay
    orig_poll = dec.st.poll_record

    def replay(self):
        marks["r0"] = ns()
        orig_replay(self)
        marks["r1"] = ns()

    def poll(step_no, *a):
        r = orig_poll(step_no, *a)
        marks["p1"] = ns()
        return r

    torch.cuda.CUDAGraph.replay = replay
    dec.st.poll_record = poll
    rows = []
    prev_end = None
    for _ in range(a.steps):
        t0 = ns()
        dec.step()
        t1 = ns()
        rows.append(dict(before=marks["r0"] - t0, launch=marks["r1"] - marks["r0"], poll=marks["p1"] - marks["r1"], after=t1 - marks["p1"],
                         loop=(t0 - prev_end) if prev_end else 0, step=t1 - t0))
        prev_end = t1
    torch.cuda.CUDAGraph.replay = orig_replay
    med = {k: statistics.median(r[k] for r in rows) / 1e3 for k in rows[0]}
    print("us per steady step (median of %d): python before replay %.1f | replay call %.1f | poll (GPU runs) %.1f | python after the record %.1f | "
          "loop %.1f | step %.1f" % (len(rows), med["before"], med["launch"], med["poll"], med["after"], med["loop"], med["step"]))
    print("host on the critical path (before + replay call + after) = %.1f us of %.1f" % (med["before"] + med["launch"] + med["after"], med["step"]))


if __name__ == "__main__":
    main()
