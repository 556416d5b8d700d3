"""Public configuration / patching API - the same surface as the reference's `lade/utils.py`.

config_lade (lade/utils.py:13-37), augment_llama / augment_generate / augment_all (:57-71),
log_history (:74-83), save_log (:85-87).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .decoding import CONFIG_MAP, FUNC_MAP


def config_lade(WINDOW_SIZE=None, LEVEL=None, DEBUG=None, GUESS_SET_SIZE=None, ALWAYS_FWD_ONE=None, SPLIT_FLAG=None,
                DIST_WORKERS=None, POOL_FROM_PROMPT=None, backend="nccl", USE_FLASH=None):
    """Same arguments and semantics as the reference.  `backend='nccl'` is RCCL on ROCm.  USE_FLASH is accepted and
    ignored: there is a single attention path (the fused HIP kernel)."""
    if WINDOW_SIZE is not None:
        CONFIG_MAP["WINDOW_SIZE"] = WINDOW_SIZE
    if LEVEL is not None:
        CONFIG_MAP["LEVEL"] = LEVEL
    if GUESS_SET_SIZE is not None:
        CONFIG_MAP["GUESS_SET_SIZE"] = GUESS_SET_SIZE
    if ALWAYS_FWD_ONE is not None:
        CONFIG_MAP["ALWAYS_FWD_ONE"] = ALWAYS_FWD_ONE
    if DEBUG is not None:
        CONFIG_MAP["DEBUG"] = DEBUG
    if SPLIT_FLAG is not None:
        CONFIG_MAP["SPLIT_FLAG"] = SPLIT_FLAG
    if POOL_FROM_PROMPT is not None:
        CONFIG_MAP["POOL_FROM_PROMPT"] = POOL_FROM_PROMPT
    if DIST_WORKERS is not None and DIST_WORKERS > 1:
        CONFIG_MAP["DIST_WORKERS"] = DIST_WORKERS
        CONFIG_MAP["LOCAL_RANK"] = int(os.environ["LOCAL_RANK"])
        if not dist.is_initialized():
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if backend == "nccl":
                torch.cuda.set_device(CONFIG_MAP["LOCAL_RANK"])
                dist.init_process_group(backend, rank=CONFIG_MAP["LOCAL_RANK"], device_id=torch.device("cuda", CONFIG_MAP["LOCAL_RANK"]))
            else:
                dist.init_process_group(backend, rank=CONFIG_MAP["LOCAL_RANK"])
        assert dist.get_world_size() == DIST_WORKERS, "DIST_WORKERS config should be equal to work size"
    if USE_FLASH is not None:
        CONFIG_MAP["USE_FLASH"] = USE_FLASH
    CONFIG_MAP["log"] = []            # the reference resets the log on every call (lade/utils.py:37)


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
    gen = 0
    step = 0
    if "log" in CONFIG_MAP:
        for log in CONFIG_MAP["log"]:
            gen += log[0]
            step += log[1]
    if clear:
        CONFIG_MAP["log"] = []
    print("LADE LOG - OVERALL GEN: ", gen, " STEPS: ", step, " AVG COMPRESS RATIO: ", (gen / step) if step > 0 else 0)


def save_log(log_dir):
    if "log" in CONFIG_MAP:
        torch.save(CONFIG_MAP["log"], log_dir)
