"""lookaheaddecoding_amd - MI355X-native lookahead decoding behind the reference's `lade` surface.

    import lookaheaddecoding_amd as lade        # or: import lade  (alias package at the repo root)
    lade.augment_all()
    lade.config_lade(LEVEL=5, WINDOW_SIZE=15, GUESS_SET_SIZE=15, DEBUG=1)
    # USE_LADE=1 python app.py  ->  model.generate(...) runs the HIP lookahead loop

Same names as `lade/__init__.py:1-5` of the reference.  The compute path is liblade_hip.so (HIP, gfx950); it is
loaded lazily, and everything that computes fails loudly when it is missing.
"""
from .utils import augment_all, augment_generate, augment_llama, config_lade, log_history, save_log  # noqa: F401
from .lade_distributed import distributed, get_device  # noqa: F401
from . import decoding  # noqa: F401  (lade.decoding.CONFIG_MAP is read by applications)

__all__ = ["augment_all", "augment_generate", "augment_llama", "config_lade", "log_history", "save_log", "get_device",
           "distributed", "decoding"]
