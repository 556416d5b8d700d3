"""Lookahead parallelism on the HIP kernels.  Only one GPU is available per test box, so the R ranks
run as R threads of one process on cuda:0 (each with its own engine, window, pool and KV cache) and the
collectives are swapped for an in-process exchange; the rank-local kernels (sharded input assembly,
lade_lp_pack, lade_lp_reduce_apply) and the orchestration are the product's.  Checked against the
reference's own gloo runs: tokens, step count and every rank's step inputs.  A real 1-rank RCCL group
is exercised as well."""
import json
import os
import random
import threading

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from conftest import GOLDEN
from lookaheaddecoding_amd.weights import make_config, random_weights_numpy


class ThreadExchange:
    def __init__(self, R):
        self.R = R
        self.bar = threading.Barrier(R)
        self.slots = [None] * R
        self.lock = threading.Lock()

    def all_gather(self, rank, out, inp):
        torch.cuda.current_stream().synchronize()
        self.slots[rank] = inp.clone()
        self.bar.wait()
        out.copy_(torch.cat([s.to(out.device) for s in self.slots]))
        torch.cuda.current_stream().synchronize()
        self.bar.wait()

    def broadcast(self, rank, t):
        if rank == 0:
            self.slots[0] = t.clone()
        self.bar.wait()
        t.copy_(self.slots[0])
        self.bar.wait()


def _run_rank(rank, R, run, ex, results, errors):
    try:
        from lookaheaddecoding_amd import parallel
        from lookaheaddecoding_amd.decoding import LookaheadDecoder
        from lookaheaddecoding_amd.engine import StepEngine
        torch.cuda.set_device(0)
        cfg = make_config(run["model"], max_pos=512)
        w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=run["model_seed"], std=run["std"]).items()}
        eng = StepEngine(cfg, w, dtype=torch.float32, max_seq=512, max_T=320)
        dec = LookaheadDecoder(eng, run["W"], run["N"], run["G"], lp=parallel.LPContext(rank=rank, world=R),
                               pool_from_prompt=bool(run.get("pool_from_prompt", 0)))
        be = parallel.HipLPBackend(dec)
        be.broadcast_window = lambda w0, lp: (lambda t: (ex.broadcast(rank, t), t.tolist())[1])(torch.tensor(w0, dtype=torch.int32, device="cuda"))
        ids_per_step = []
        orig = be.local_step

        def rec_step(phase, P, n_input, level_lens, c0, c1, g, glo, ghi):
            r = orig(phase, P, n_input, level_lens, c0, c1, g, glo, ghi)
            ls = parallel.shard_level_sizes(level_lens, c0, c1)
            T = (len(run["prompt"]) + c1 - 1) if phase == 0 else n_input + sum(ls) + (ghi - glo) * (run["N"] - 1)
            ids_per_step.append(dec.st.ids[:T].cpu().tolist() if phase != 0 else None)
            return r

        be.local_step = rec_step
        with torch.cuda.stream(torch.cuda.Stream()):
            out = _greedy_lp_patched(parallel, dec, run, be, ex, rank)
        results[rank] = (out.tokens, out.steps, ids_per_step)
    except Exception as e:  # pragma: no cover
        import traceback
        errors.append((rank, traceback.format_exc()))
        try:
            ex.bar.abort()
        except Exception:
            pass


def _greedy_lp_patched(parallel, dec, run, be, ex, rank):
    return parallel.greedy_lp(dec, run["prompt"], run["max_length"], eos_token_id=run.get("eos"), rng=random.Random(run["seed"] + 1000 * rank), backend=be, keep_trace=True,
                              all_gather=lambda out, inp: ex.all_gather(rank, out, inp))


@pytest.mark.parametrize("idx", [0, 1, 2, 3, 4, 5, 6, 7])
def test_lp_hip_kernels_match_reference_gloo_runs(idx):
    with open(os.path.join(GOLDEN, "e2e_lp.json")) as f:
        run = json.load(f)["runs"][idx]
    R = run["R"]
    torch.cuda.init()
    torch.zeros(1, device="cuda")
    ex = ThreadExchange(R)
    results, errors = {}, []
    threads = [threading.Thread(target=_run_rank, args=(r, R, run, ex, results, errors)) for r in range(R)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors[0][1]
    for rank in range(R):
        toks, steps, ids = results[rank]
        assert toks == run["tokens"], (rank, R)
        assert steps == run["steps"]
        for i, (mine, ref) in enumerate(zip(ids, run["rank_traces"][rank])):
            if mine is not None:
                assert mine == ref["ids"], (rank, i)


def test_lp_single_rank_over_rccl():
    """world_size 1 through the real RCCL backend: tokens and step count of the reference."""
    from lookaheaddecoding_amd import parallel
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    with open(os.path.join(GOLDEN, "e2e_greedy.json")) as f:
        run = json.load(f)["runs"][0]
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cfg = make_config(run["model"], max_pos=512)
        w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=run["model_seed"], std=run["std"]).items()}
        eng = StepEngine(cfg, w, dtype=torch.float32, max_seq=512, max_T=320)
        dec = LookaheadDecoder(eng, run["W"], run["N"], run["G"], lp=parallel.LPContext(rank=0, world=1))
        out = parallel.greedy_lp(dec, run["prompt"], run["max_length"], rng=random.Random(run["seed"]))
        assert out.tokens == run["tokens"] and out.steps == run["steps"]
    finally:
        dist.destroy_process_group()


def test_lp_collective_through_the_c_abi_single_rank():
    """lade_lp_unique_id / lade_lp_comm_create / lade_lp_allgather / lade_lp_comm_destroy on a real 1-rank RCCL communicator, and a
    whole greedy run whose per-step collective goes through it (LADE_LP_COLLECTIVE=abi): tokens and step count of the reference."""
    from lookaheaddecoding_amd import parallel
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    torch.cuda.set_device(0)
    comm = parallel.RcclComm(0, 1)
    src = torch.arange(37, dtype=torch.int32, device="cuda") * 3 + 1
    dst = torch.zeros(37, dtype=torch.int32, device="cuda")
    comm.all_gather(dst, src)
    torch.cuda.synchronize()
    assert torch.equal(dst, src)
    g = torch.cuda.CUDAGraph()                      # the collective is ordered on the step's stream and can be captured with it
    src2 = src.clone()
    with torch.cuda.graph(g):
        comm.all_gather(dst, src2)
    src2 += 5
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(dst, src + 5)
    comm.close()
    with open(os.path.join(GOLDEN, "e2e_greedy.json")) as f:
        run = json.load(f)["runs"][1]
    cfg = make_config(run["model"], max_pos=512)
    w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=run["model_seed"], std=run["std"]).items()}
    eng = StepEngine(cfg, w, dtype=torch.float32, max_seq=512, max_T=320)
    dec = LookaheadDecoder(eng, run["W"], run["N"], run["G"], lp=parallel.LPContext(rank=0, world=1), pool_from_prompt=bool(run["pool_from_prompt"]))
    os.environ["LADE_LP_COLLECTIVE"] = "abi"
    try:
        out = parallel.greedy_lp(dec, run["prompt"], run["max_length"], eos_token_id=run["eos"], rng=random.Random(run["seed"]))
    finally:
        os.environ.pop("LADE_LP_COLLECTIVE", None)
    assert out.tokens == run["tokens"] and out.steps == run["steps"]


def _lp_graph_threads(run, R):
    from lookaheaddecoding_amd import parallel
    from lookaheaddecoding_amd.decoding import LookaheadDecoder
    from lookaheaddecoding_amd.engine import StepEngine
    torch.zeros(1, device="cuda")
    ex = ThreadExchange(R)
    results, errors = {}, []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            cfg = make_config(run["model"], max_pos=512)
            w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=run["model_seed"], std=run["std"]).items()}
            eng = StepEngine(cfg, w, dtype=torch.float32, max_seq=512, max_T=320)
            dec = LookaheadDecoder(eng, run["W"], run["N"], run["G"], lp=parallel.LPContext(rank=rank, world=R),
                                   pool_from_prompt=bool(run.get("pool_from_prompt", 0)), use_graph=True)
            be = parallel.HipLPBackend(dec)
            be.broadcast_window = lambda w0, lp: (lambda t: (ex.broadcast(rank, t), t.tolist())[1])(torch.tensor(w0, dtype=torch.int32, device="cuda"))
            with torch.cuda.stream(torch.cuda.Stream()):
                out = parallel.greedy_lp(dec, run["prompt"], run["max_length"], eos_token_id=run.get("eos"), rng=random.Random(run["seed"] + 1000 * rank),
                                         backend=be, all_gather=lambda o, i: ex.all_gather(rank, o, i))
            results[rank] = (out.tokens, out.steps, len(getattr(be, "_segments", {})))
        except Exception:  # pragma: no cover
            import traceback
            errors.append((rank, traceback.format_exc()))
            try:
                ex.bar.abort()
            except Exception:
                pass

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(R)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors[0][1]
    return results


@pytest.mark.parametrize("idx", [0, 2, 5])
def test_lp_steady_steps_as_graph_segments_match_reference_gloo_runs(idx):
    """use_graph under lookahead parallelism: the rank-local part of a steady step (sharded input assembly with the candidate share
    decided on the device, forward, argmax, record) replays as a hipGraph segment per (re-fed inputs, candidate bucket); tokens and
    step counts of the reference's gloo runs, including runs whose hits are re-fed (n_input > 1) and POOL_FROM_PROMPT."""
    with open(os.path.join(GOLDEN, "e2e_lp.json")) as f:
        run = json.load(f)["runs"][idx]
    res = _lp_graph_threads(run, run["R"])
    for rank in range(run["R"]):
        toks, steps, n_seg = res[rank]
        assert toks == run["tokens"] and steps == run["steps"], (rank, idx)
        assert n_seg >= 1, "no graph segment was captured"


def test_lp_graph_segments_single_rank_equals_single_gpu_runs():
    with open(os.path.join(GOLDEN, "e2e_greedy.json")) as f:
        runs = json.load(f)["runs"]
    for run in runs[:4]:
        res = _lp_graph_threads(run, 1)
        assert res[0][0] == run["tokens"] and res[0][1] == run["steps"]


def test_lp_rank_whose_graph_segment_fails_falls_back_to_eager_steps_and_keeps_the_stream(monkeypatch, capfd):
    """A hipGraph segment that cannot be captured or replayed (the path has never run next to a multi-GPU RCCL communicator) must not end
    the run: the rank restores the integer state the warm-up touched, takes eager lookahead-parallel steps from there on and the
    tokens / steps stay those of the reference's run.  Forced here by a forward() that fails once, inside the segment's warm-up."""
    from lookaheaddecoding_amd.engine import StepEngine
    with open(os.path.join(GOLDEN, "e2e_greedy.json")) as f:
        run = json.load(f)["runs"][1]
    orig = StepEngine.forward
    state = {"armed": True}

    def flaky(self, *a, **k):
        if state["armed"] and k.get("dyn_P") is not None:          # the first device-length forward = the first segment's warm-up
            state["armed"] = False
            raise RuntimeError("simulated capture failure")
        return orig(self, *a, **k)

    monkeypatch.setattr(StepEngine, "forward", flaky)
    res = _lp_graph_threads(run, 1)
    assert not state["armed"], "the failure was never injected"
    assert res[0][0] == run["tokens"] and res[0][1] == run["steps"]
    assert "eager lookahead-parallel steps from here on" in capfd.readouterr().err


def test_lp_loop_on_the_c_abi_communicator_without_torch_distributed(tmp_path):
    """A caller without torch.distributed: RcclComm built over a byte channel (here a world of one rank, so none is needed), handed to the
    package's lookahead-parallel loop - the step's all-gather, the window broadcast and the GEMM-table adoption all go through
    lade_lp_allgather.  Run in a fresh process that never initialises torch.distributed; tokens == single-GPU decoding."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r"""
import random, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from lookaheaddecoding_amd import parallel
from lookaheaddecoding_amd.decoding import LookaheadDecoder
from lookaheaddecoding_amd.engine import StepEngine
from lookaheaddecoding_amd.weights import make_config, random_weights_numpy
cfg = make_config("tiny-d128", max_pos=512)
w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg, seed=2, std=0.08).items()}
prompt = [1, 5, 9, 17, 33, 5, 9, 17, 44, 5, 9, 17, 33, 5, 9, 17]
for dt in (torch.float32, torch.bfloat16):
    eng = StepEngine(cfg, w, dtype=dt, max_seq=512, max_T=320)
    single = LookaheadDecoder(eng, 7, 4, 7, pool_from_prompt=True).greedy(prompt, len(prompt) + 40, rng=random.Random(3))
    comm = parallel.RcclComm(0, 1)
    assert comm.count() == 1
    dec = LookaheadDecoder(eng, 7, 4, 7, pool_from_prompt=True, lp=parallel.LPContext(rank=0, world=1, force=True), use_graph=True)
    out = parallel.greedy_lp(dec, prompt, len(prompt) + 40, rng=random.Random(3), comm=comm)
    comm.close()
    assert out.tokens == single.tokens and out.steps == single.steps, (dt, out.tokens, single.tokens)
assert not dist.is_initialized()
print("OK")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-1500:] + "\n" + r.stderr[-3000:]
