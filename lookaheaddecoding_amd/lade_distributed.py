"""Rank helpers of the lookahead-parallel mode, same two names as the reference exports from `lade`
(lade/lade_distributed.py:5-12; `from .lade_distributed import *` in lade/__init__.py).

Both read the shared configuration that `config_lade(DIST_WORKERS=...)` fills in; one process drives one MI355X."""
from .decoding import CONFIG_MAP

__all__ = ["get_device", "distributed"]


def get_device() -> int:
    """Index of the GPU this process owns: its LOCAL_RANK once lookahead parallelism is configured, GPU 0 before that."""
    return CONFIG_MAP.get("LOCAL_RANK", 0)


def distributed() -> bool:
    """True when the W window columns are sharded over more than one rank (DIST_WORKERS > 1)."""
    return CONFIG_MAP.get("DIST_WORKERS", 1) > 1
