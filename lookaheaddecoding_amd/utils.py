"""Public configuration / patching API - the same surface as the reference's `lade/utils.py`.

config_lade (lade/utils.py:13-37), augment_llama / augment_generate / augment_all (:57-71),
log_history (:74-83), save_log (:85-87).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .decoding import CONFIG_MAP, FUNC_MAP


# plain settings that config_lade copies into CONFIG_MAP when they are given (None = leave as is)
_PLAIN_SETTINGS = ("WINDOW_SIZE", "LEVEL", "GUESS_SET_SIZE", "ALWAYS_FWD_ONE", "DEBUG", "SPLIT_FLAG", "POOL_FROM_PROMPT", "USE_FLASH")


def config_lade(WINDOW_SIZE=None, LEVEL=None, DEBUG=None, GUESS_SET_SIZE=None, ALWAYS_FWD_ONE=None, SPLIT_FLAG=None,
                DIST_WORKERS=None, POOL_FROM_PROMPT=None, backend="nccl", USE_FLASH=None):
    """Same signature and semantics as the reference's `lade.config_lade` (lade/utils.py:13-37): every argument that is not
    None overwrites its entry of `lade.decoding.CONFIG_MAP`, `DIST_WORKERS > 1` joins the process group (LOCAL_RANK from the
    environment; `backend='nccl'` is RCCL on ROCm) and every call starts a fresh log.  USE_FLASH is stored but has no
    effect: there is a single attention path, the fused HIP kernel."""
    given = dict(WINDOW_SIZE=WINDOW_SIZE, LEVEL=LEVEL, GUESS_SET_SIZE=GUESS_SET_SIZE, ALWAYS_FWD_ONE=ALWAYS_FWD_ONE, DEBUG=DEBUG,
                 SPLIT_FLAG=SPLIT_FLAG, POOL_FROM_PROMPT=POOL_FROM_PROMPT, USE_FLASH=USE_FLASH)
    CONFIG_MAP.update({k: given[k] for k in _PLAIN_SETTINGS if given[k] is not None})
    if DIST_WORKERS is not None and DIST_WORKERS > 1:
        _join_lookahead_parallel_group(int(DIST_WORKERS), backend)
    CONFIG_MAP["log"] = []            # the reference resets the log on every call (lade/utils.py:37)


def _join_lookahead_parallel_group(workers: int, backend: str) -> None:
    """One process per GPU: rank = LOCAL_RANK (single node, as in the reference, lade/utils.py:28-35)."""
    rank = int(os.environ["LOCAL_RANK"])
    CONFIG_MAP["DIST_WORKERS"], CONFIG_MAP["LOCAL_RANK"] = workers, rank
    if not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            torch.cuda.set_device(rank)
            dist.init_process_group(backend, rank=rank, device_id=torch.device("cuda", rank))
        else:
            dist.init_process_group(backend, rank=rank)
            if torch.cuda.is_available() and rank < torch.cuda.device_count():
                torch.cuda.set_device(rank)          # the reference binds rank r to GPU r whatever the backend (lade/utils.py:31)
    assert dist.get_world_size() == workers, "DIST_WORKERS config should be equal to work size"


def augment_llama():
    """The reference copies its modified Llama methods over HF's classes (lade/utils.py:57-58).  Here the model step
    is the HIP step engine, attached lazily to each LlamaForCausalLM at its first step; `jforward_multilevel` (the
    model-step boundary the decode loop calls, lade/models/modeling_llama.py:1381) is installed on HF's class with the
    reference's signature.  Loading the HIP extension here makes a broken install fail at augment time."""
    from . import cabi, hf
    cabi.load_library()
    from transformers.models.llama import modeling_llama
    modeling_llama.LlamaForCausalLM.jforward_multilevel = hf.jforward_multilevel


def augment_generate():
    """Swaps the decode loops of `GenerationMixin` for the env-gated proxies (lade/utils.py:62-66)."""
    from transformers import GenerationMixin
    from . import hf
    if hasattr(GenerationMixin, "greedy_search"):                 # transformers <= 4.3x
        FUNC_MAP.setdefault("greedy_search", GenerationMixin.greedy_search)
        FUNC_MAP.setdefault("sample", GenerationMixin.sample)
        GenerationMixin.greedy_search = hf.greedy_search_proxy
        GenerationMixin.sample = hf.sample_proxy
    if hasattr(GenerationMixin, "_sample"):                       # transformers >= 4.4x / 5.x
        if GenerationMixin._sample is not hf._sample_proxy:
            FUNC_MAP["_sample"] = GenerationMixin._sample
            GenerationMixin._sample = hf._sample_proxy


def augment_all():
    augment_llama()
    augment_generate()


def log_history(clear=False):
    """Prints the totals of the per-generate log entries [generated, steps, ratio] (lade/utils.py:74-83)."""
    entries = CONFIG_MAP.get("log", [])
    gen, step = sum(e[0] for e in entries), sum(e[1] for e in entries)
    if clear:
        CONFIG_MAP["log"] = []
    print("LADE LOG - OVERALL GEN: ", gen, " STEPS: ", step, " AVG COMPRESS RATIO: ", (gen / step) if step > 0 else 0)


def save_log(log_dir):
    """torch.save of the log list to `log_dir` (lade/utils.py:85-87)."""
    if "log" in CONFIG_MAP:
        torch.save(CONFIG_MAP["log"], log_dir)
