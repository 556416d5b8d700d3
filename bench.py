#!/usr/bin/env python
"""Benchmark of the lookahead-decoding hot path on MI355X.

Metric (BASELINE.json): tokens/s + step-compression of greedy lookahead decoding on a synthetic
random-weight Llama-2-7B-shaped model in bf16, W=15 N=5 G=15 (BASELINE config 2), 1/2/4/8 GPUs.

A "step" is one decode step of the lookahead loop = one model forward over T=(N-1)(W+g) tokens
through the HIP hot path (input assembly, RoPE+KV append, lookahead attention, argmax, verify, pool
insert, window roll, KV commit) with the weights, KV cache, window and n-gram pool resident in HBM.
The prompt prefill and the N-2 window-fill steps are setup and are not timed; W warm-up steps and
exactly K timed steps follow, bracketed by barrier + torch.cuda.synchronize().

    python bench.py --gpus 1 --steps 32 --warmup 8
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

With N > 1 the step runs lookahead-parallel (window columns + candidates sharded over the ranks, one
RCCL all-gather of a small int32 record per step; lade/decoding.py:973-986, 1088-1107): total work
per step is fixed, so scaling is "strong".

Prints ONE JSON line (rank 0) with the driver's contract plus `roofline` (the attention kernel,
algorithmic bytes / live hipEvent timing) and `cpu_baseline` (the CPU oracle = a port of the
reference's greedy path, timed on this box's host cores on a bounded layer-sliced sample).
"""
from __future__ import annotations

import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="llama2-7b")
    ap.add_argument("--layers", type=int, default=0, help="override the layer count (debug only; invalidates the metric)")
    ap.add_argument("--window", type=int, default=15)
    ap.add_argument("--level", type=int, default=5)
    ap.add_argument("--guess", type=int, default=15)
    ap.add_argument("--prompt-len", type=int, default=2048)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--no-graph", action="store_true", help="run steady steps eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--force-lp", action="store_true", help="run the lookahead-parallel code path even with one rank (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", type=int, default=3)
    return ap.parse_args()


def attn_algorithmic_bytes(cfg, T, P, elem=2):
    """K1 bytes per layer = e*[2*Hkv*(P+T)*d (K,V read) + H*T*d (Q read) + H*T*d (O write)]"""
    H, Hkv, d = cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
    return elem * (2 * Hkv * (P + T) * d + 2 * H * T * d)


def pmc_traffic(T, P, n_splits):
    """HBM bytes per launch pair from the rocprofv3 PMC passes committed under profiles/ (bench.py cannot collect
    counters itself): FETCH_SIZE (x2 on gfx950 for wide coalesced reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE,
    KiB -> bytes, attention + combine kernels.  Only reported when the profiled shape matches this run's shape."""
    path = os.path.join(ROOT, "profiles", "r1_attn_pmc.json")
    try:
        with open(path) as f:
            for e in json.load(f)["entries"]:
                if e["T"] == T and abs(e["P"] - P) <= 64 and e["n_splits"] == n_splits:
                    return e["traffic_bytes"]
    except Exception:
        pass
    return None


def host_cores() -> int:
    """cores this process may actually use: the cgroup CPU quota when there is one, else the affinity mask"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(args, cfg_full):
    """The reference's CPU greedy path cannot travel to this box; its port (oracle/lade_oracle.py, pinned
    to reference-generated traces) is timed instead, fp32, all host cores, on a bounded sample: a
    layer-sliced model of the same widths, short prompt, a few steady steps; the per-layer time is
    extrapolated to the full depth."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import lade_oracle as O
    from lookaheaddecoding_amd.weights import make_config, weight_shapes
    cores = host_cores()
    torch.set_num_threads(cores)
    W, N, G = args.window, args.level, args.guess
    prompt_len = 64
    times = {}
    for Ls in (1, 2):
        cfg = make_config(cfg_full, layers=Ls, max_pos=1024)
        g = torch.Generator().manual_seed(0)
        w = {}
        for k, shp in weight_shapes(cfg).items():
            w[k] = torch.ones(shp) if len(shp) == 1 else torch.empty(shp).normal_(0, 0.02, generator=g)
        model = O.OracleLlama(cfg, w)
        prompt = torch.randint(3, cfg["vocab"], (prompt_len,), generator=torch.Generator().manual_seed(123)).tolist()
        n_steps = (N - 1) + args.cpu_baseline_steps
        # time whole steps of the oracle loop; the last `cpu_baseline_steps` are steady steps
        t_marks = []
        orig = O.model_step

        def timed_step(*a, **k):
            t0 = time.time()
            r = orig(*a, **k)
            t_marks.append((time.time() - t0, r.layout.T))
            return r

        O.model_step = timed_step
        try:
            res = O.lookahead_greedy(model, prompt, W, N, G, prompt_len + n_steps, random.Random(1), keep_trace=False)
        finally:
            O.model_step = orig
        steady = t_marks[N - 1:]
        times[Ls] = (sum(t for t, _ in steady) / max(1, len(steady)), sum(T for _, T in steady) / max(1, len(steady)), res.steps)
        del model, w
    per_layer = max(times[2][0] - times[1][0], 1e-9)
    fixed = max(times[1][0] - per_layer, 0.0)
    step_s = fixed + cfg_full["layers"] * per_layer
    return {"value": round(1.0 / step_s, 4), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle/lade_oracle.py (port of lade/decoding.py:697-1259 + modeling_llama.py eager path), fp32, {cores} threads; "
                      f"layer-sliced {args.model} shape (1 and 2 layers -> per-layer {per_layer * 1e3:.1f} ms, fixed {fixed * 1e3:.1f} ms, "
                      f"extrapolated to {cfg_full['layers']} layers = {step_s:.2f} s/step), prompt {prompt_len}, {args.cpu_baseline_steps} steady steps, "
                      f"W={W} N={N} G={G}, T~{times[2][1]:.0f} tokens/step, S=1.0 (cold regime: 1 token/step)",
            "s_per_step": round(step_s, 4)}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_lp = world > 1 or args.force_lp
    if use_lp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from lookaheaddecoding_amd import ops
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    from lookaheaddecoding_amd.weights import make_config, random_weights_torch

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    cfg = make_config(args.model)
    if args.layers:
        cfg["layers"] = args.layers
    W, N, G = args.window, args.level, args.guess
    gs = N - 1
    total_steps = (N - 1) + args.warmup + args.steps + 2
    max_seq = args.prompt_len + total_steps * N + (N - 1) * (W + G) + 64
    cfg["max_pos"] = max(cfg.get("max_pos", 4096), max_seq)
    weights = random_weights_torch(cfg, seed=0, dtype=dtype, device=dev)
    eng = StepEngine(cfg, weights, dtype=dtype, device=dev, max_seq=max_seq, max_T=512)
    del weights
    lp = None
    if use_lp:
        from lookaheaddecoding_amd.parallel import LPContext
        lp = LPContext(rank=rank, world=world)
    dec = LookaheadDecoder(eng, W, N, G, lp=lp, use_graph=not args.no_graph and not use_lp)
    prompt = torch.randint(3, cfg["vocab"], (args.prompt_len,), generator=torch.Generator().manual_seed(123)).tolist()

    def sync():
        if use_lp:
            dist.barrier()
        torch.cuda.synchronize()

    run = dec
    if use_lp:
        from lookaheaddecoding_amd.parallel import LPRunner
        run = LPRunner(dec)
    run.start(prompt, rng=random.Random(1))
    for _ in range(N - 1):                       # prefill + window fill: setup, untimed
        run.step()
    for _ in range(args.warmup):
        run.step()
    sync()
    tok0 = len(run.tokens)
    t0 = time.perf_counter()
    infos = []
    for _ in range(args.steps):
        infos.append(run.step())
    sync()
    elapsed = time.perf_counter() - t0
    if use_lp:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    new_tokens = len(run.tokens) - tok0
    S = new_tokens / args.steps
    Ts = [i["T"] for i in infos if i.get("T")]
    avg_T = (sum(Ts) / len(Ts)) if Ts else float((N - 1) * W)
    P_end = run.P

    # ---- the attention launch pair timed IN the step: a few more steady steps, eager, with a hipEvent before the
    # attention launch and after the split merge of every layer (torch's current stream is the launch stream) ----
    in_situ = None
    if rank == 0 and not use_lp:
        was_graph, run.use_graph = run.use_graph, False
        run.step()                                   # first eager step: one-off costs (autotune of a new row class ...)
        sync()
        eng.attn_events = []
        te0 = time.perf_counter()
        for _ in range(4):
            run.step()
        sync()
        eager_ms = (time.perf_counter() - te0) / 4 * 1e3
        evs, eng.attn_events, run.use_graph = eng.attn_events, None, was_graph
        T_ref = int(round(avg_T))
        evs = [e for e in evs if e[2] == T_ref] or evs           # launches of the steady shape only (a stray candidate changes T)
        durs = sorted(e0.elapsed_time(e1) * 1e3 for (e0, e1, _, _) in evs)
        if durs:
            # eager launches keep the GPU busy only when a step's GPU time exceeds its host launch time; otherwise the
            # bracket also contains host gaps (small models) and the isolated timing is reported instead
            gpu_bound = eager_ms <= 1.15 * (elapsed / args.steps * 1e3)
            in_situ = {"us": sum(durs) / len(durs), "median_us": durs[len(durs) // 2], "T": evs[-1][2], "n_splits": evs[-1][3], "P": run.P,
                       "launches": len(durs), "eager_ms_per_step": round(eager_ms, 3), "gpu_bound": bool(gpu_bound)}

    # ---- the same pair as a step-time difference: a second decoder over the same engine with the attention launches left
    # out, same hipGraph mode, same shapes (random weights: no candidates either way) - independent of the host's launch rate
    graph_delta = None
    if rank == 0 and not use_lp and not args.no_graph:
        eng.skip_attn = True
        d2 = LookaheadDecoder(eng, W, N, 0, use_graph=True)      # no candidates: the garbage logits must not change the step shape
        d2.start(prompt, rng=random.Random(1))
        for _ in range(N - 1 + args.warmup):
            d2.step()
        sync()
        tg0 = time.perf_counter()
        i2 = [d2.step() for _ in range(args.steps)]
        sync()
        ms_noattn = (time.perf_counter() - tg0) / args.steps * 1e3
        eng.skip_attn = False
        T2 = sum(i["T"] for i in i2) / len(i2)
        same_class = lambda t: (t <= 32, t <= 64, t <= 96, t <= 128)
        if abs(T2 - avg_T) <= 4 and all(same_class(i["T"]) == same_class(int(round(avg_T))) for i in i2):      # same GEMM row class throughout
            graph_delta = {"us": (elapsed / args.steps * 1e3 - ms_noattn) / cfg["layers"] * 1e3, "ms_per_step_without_attention": round(ms_noattn, 3)}

    # ---- plain autoregressive decoding on the same engine and cache length (one token per forward, T = 1): what
    # lookahead decoding has to beat; S * (plain step / lookahead step) is its speed-up
    plain = None
    if rank == 0 and not use_lp:
        one_id = torch.full((1,), 5, dtype=torch.int32, device=dev)
        one_pos = torch.full((1,), P_end, dtype=torch.int32, device=dev)
        sel0 = torch.zeros(1, dtype=torch.int32, device=dev)
        m1 = ops.StepMask(T=1, P=P_end, is_prefill=True)

        def plain_step():
            ops.argmax_rows(eng.forward(one_id, one_pos, m1, sel0, 1))

        for _ in range(3):
            plain_step()
        gph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gph):
            plain_step()
        sync()
        tp0 = time.perf_counter()
        for _ in range(args.steps):
            gph.replay()
        sync()
        ms_plain = (time.perf_counter() - tp0) / args.steps * 1e3
        plain = {"value": round(1e3 / ms_plain, 2), "unit": "tokens/s", "ms_per_token": round(ms_plain, 3),
                 "how": f"one-token forward + argmax as a hipGraph at cache length {P_end}, same engine and kernels"}

    # ---- hot regime, measured last because it overwrites weights.  Random weights never accept a
    # candidate (S = 1).  To time the accept path under load the model is turned into a deterministic successor map:
    # every layer's o_proj / down_proj zeroed (the residual stream keeps the input embedding) and lm_head row j set to
    # embed[(j-1) mod C] for j < C, so the greedy continuation of token t is (t+1) mod C; the prompt walks that cycle
    # and POOL_FROM_PROMPT seeds the pool, so every step verifies a full n-gram (S -> N-1).  Same kernels, same bytes.
    def hot_regime():
        if use_lp:
            return None
        C = 256
        for lw in eng.layers:
            lw["wo"].zero_()
            lw["wd"].zero_()
        head = eng.embed.clone()
        head[:C] = eng.embed[(torch.arange(C, device=dev) - 1) % C]
        eng.lm_head = head
        hot_dec = LookaheadDecoder(eng, W, N, G, pool_from_prompt=True, use_graph=not args.no_graph)
        hot_prompt = [i % C for i in range(args.prompt_len)]
        hot_dec.start(hot_prompt, rng=random.Random(1))
        for _ in range(N - 1 + args.warmup):
            hot_dec.step()
        sync()
        tok_h = len(hot_dec.tokens)
        th0 = time.perf_counter()
        hot_infos = [hot_dec.step() for _ in range(args.steps)]
        sync()
        th = time.perf_counter() - th0
        gen = hot_dec.tokens[tok_h:]
        ok = all(gen[i + 1] == (gen[i] + 1) % C for i in range(len(gen) - 1))
        return {"value": round(len(gen) / th, 2), "unit": "tokens/s", "step_compression": round(len(gen) / args.steps, 3),
                "ms_per_step": round(th / args.steps * 1e3, 3), "tokens_per_step_T": round(sum(i["T"] for i in hot_infos) / len(hot_infos), 1),
                "output_is_the_successor_cycle": ok,
                "how": "successor-map model (o_proj/down_proj zeroed, lm_head = shifted embedding), cyclic prompt, POOL_FROM_PROMPT=1"}

    out = None
    if rank == 0:
        # ---- roofline of the dominant hand-written kernel: lookahead attention, one layer ----
        T_mid = int(round(avg_T))
        g_mid = max(0, (T_mid - (N - 1) * W) // gs)
        T_k = (N - 1) * W + g_mid * gs
        mask = ops.StepMask.from_levels(1, [W - 1] + [W] * (N - 2), g_mid * gs, gs, P_end)
        qkv = torch.randn(T_k, (cfg["heads"] + 2 * cfg["kv_heads"]) * cfg["head_dim"], device=dev).to(dtype)
        ns = eng.n_splits_for(T_k, P_end + T_k)
        # every repetition uses the next layer's K/V cache (L caches of 2*Hkv*S_max*d*e bytes >> the 256 MB Infinity Cache), so
        # the launch streams its keys/values from HBM exactly as inside a decode step
        us = ops.time_attn(qkv, [eng.k_cache(li) for li in range(eng.L)], [eng.vt_cache(li) for li in range(eng.L)], mask,
                           H=cfg["heads"], Hkv=cfg["kv_heads"], d=cfg["head_dim"], n_splits=ns, reps=max(200, 8 * eng.L))
        us_iso = us
        if in_situ is not None and in_situ["T"] == T_k and in_situ["gpu_bound"]:
            us, ns = in_situ["us"], in_situ["n_splits"]
        elif graph_delta is not None and graph_delta["us"] > 0:
            us = graph_delta["us"]
        alg = attn_algorithmic_bytes(cfg, T_k, P_end)
        achieved = alg / (us * 1e-6) / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(achieved / 8000.0, 4),
                    "traffic": pmc_traffic(T_k, P_end, ns), "kernel": f"lade::attn_fwd_kernel<{args.dtype},{cfg['head_dim']}> (+combine, n_splits={ns})",
                    "launch_us": round(us, 2), "algorithmic_bytes": alg, "T": T_k, "P": P_end,
                    "launch_us_in_step": None if in_situ is None else {k: (round(v, 2) if isinstance(v, float) else v) for k, v in in_situ.items()},
                    "launch_us_graph_delta": None if graph_delta is None else {k: (round(v, 2) if isinstance(v, float) else v) for k, v in graph_delta.items()},
                    "launch_us_isolated": round(us_iso, 2),
                    "note": "launch_us = one layer's launch pair (attention + split merge) bracketed by hipEvents on the launch stream INSIDE real decode steps "
                            "(4 eager steady steps after the timed region, every layer), mean - used when those eager steps are GPU bound "
                            "(launch_us_in_step.gpu_bound), else launch_us_graph_delta = (graph step time - graph step time with the attention launches "
                            "left out) / layers; launch_us_isolated = the same pair launched back to back "
                            "(lade_time_attn_rot, cycling through the layers' K/V caches so that every launch reads HBM)"}
        hot = hot_regime()
        cpu = None
        if not args.no_cpu_baseline and world == 1:            # the CPU baseline is timed at N=1 only
            cpu = cpu_baseline(args, cfg)
        out = {
            "metric": "tokens/s, greedy lookahead decoding (W=15,N=5,G=15)" if (W, N, G) == (15, 5, 15) else f"tokens/s, greedy lookahead decoding (W={W},N={N},G={G})",
            "value": round(new_tokens / elapsed, 2), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (random-init weights, random prompt ids)",
            "config": {"workload": f"{args.model}-shape ({cfg['layers']}L) {args.dtype} greedy lookahead, 1 sequence, prompt {args.prompt_len}, "
                                   f"W={W} N={N} G={G}, cold regime (untied random weights)", "parallelism": f"lp{world}" if use_lp else "single",
                       "tokens_per_step_T": round(avg_T, 1), "kv_len_end": P_end, "hipgraph": bool(dec.use_graph)},
            "step_compression": round(S, 3), "steps_per_s": round(args.steps / elapsed, 2),
            "hot_regime": hot, "plain_decode": plain, "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if use_lp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
