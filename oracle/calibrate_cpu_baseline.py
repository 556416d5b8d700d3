"""Calibrates bench.py's `cpu_baseline` (kind "port"): the REFERENCE's own greedy lookahead loop (unmodified /root/reference, loaded through
ref_shim) and its CPU restatement (oracle/lade_oracle.py) timed on the SAME model, prompt, window RNG and cores, in this build container.

The GPU box has no /root/reference, so bench.py can only time the port there; this script states what that number is worth: the ratio
port / reference on (a) BASELINE config 1 at full depth (TinyLlama-1.1B shape, W=5 N=3 G=3: minimal.py's case) and (b) the Llama-2-7B width
at 2 layers (W=15 N=5 G=15).  Token streams must be identical (they are the pinned parity).  Writes oracle/cpu_calibration.json, which
bench.py quotes in `cpu_baseline.sample`.  TEST INFRASTRUCTURE; run in the build container only:  python oracle/calibrate_cpu_baseline.py"""
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G              # loads the reference through the shim (its module body only defines generators)
import torch

import lade_oracle as O
from lookaheaddecoding_amd.weights import make_config, weight_shapes


def weights_fp32(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    return {k: (torch.ones(s) if len(s) == 1 else torch.empty(s).normal_(0, 0.02, generator=g)) for k, s in weight_shapes(cfg).items()}


def one(tag, cfg, W, N, Gs, prompt_len, n_new, reps=2):
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    w = weights_fp32(cfg)
    prompt = torch.randint(3, cfg["vocab"], (prompt_len,), generator=torch.Generator().manual_seed(123)).tolist()
    max_length = prompt_len + n_new
    ref_model = G.build_ref_model(cfg, w)
    t_ref, t_port = [], []
    for _ in range(reps):
        t0 = time.time()
        ref_tokens, ref_steps, ref_gen, _rec = G.run_ref_greedy(ref_model, prompt, W, N, Gs, max_length, seed=1)
        t_ref.append(time.time() - t0)
    del ref_model
    model = O.OracleLlama(cfg, w)
    for _ in range(reps):
        t0 = time.time()
        out = O.lookahead_greedy(model, prompt, W, N, Gs, max_length, random.Random(1), keep_trace=False)
        t_port.append(time.time() - t0)
    assert out.tokens == ref_tokens and out.steps == ref_steps, (tag, "port and reference disagree")
    r = {"case": tag, "layers": cfg["layers"], "W": W, "N": N, "G": Gs, "prompt_len": prompt_len, "new_tokens": ref_gen, "steps": ref_steps, "cores": cores,
         "reference_s": round(min(t_ref), 3), "port_s": round(min(t_port), 3), "reference_tokens_per_s": round(ref_gen / min(t_ref), 3),
         "port_tokens_per_s": round(ref_gen / min(t_port), 3), "port_over_reference_time": round(min(t_port) / min(t_ref), 3), "tokens_identical": True}
    print(json.dumps(r), flush=True)
    return r


if __name__ == "__main__":
    res = [one("C1: TinyLlama-1.1B shape, full depth (minimal.py's configuration)", make_config("tinyllama-1.1b", max_pos=1024), 5, 3, 3, 64, 32),
           one("Llama-2-7B width, 2 layers", make_config("llama2-7b", layers=2, max_pos=1024), 15, 5, 15, 256, 16)]
    note = ("wall time of the whole generate call (prefill + steps), best of 2, fp32, torch threads = all cores of the build container; the port restates "
            "the reference's per-step work op for op (dense fp32 mask, torch.cat of the cache, lm_head over the rows the reference reads) - "
            "port_over_reference_time is what bench.py's cpu_baseline (kind 'port') must be divided by to read as the reference's own speed")
    with open(os.path.join(HERE, "cpu_calibration.json"), "w") as f:
        json.dump({"cases": res, "note": note}, f, indent=1)
