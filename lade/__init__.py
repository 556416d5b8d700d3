"""`import lade` drop-in: alias of `lookaheaddecoding_amd` (the MI355X-native implementation), so code written
against hao-ai-lab/LookaheadDecoding (`lade.augment_all()`, `lade.config_lade(...)`, `lade.decoding.CONFIG_MAP`)
runs unchanged."""
import sys

import lookaheaddecoding_amd as _impl
from lookaheaddecoding_amd import *  # noqa: F401,F403
from lookaheaddecoding_amd import decoding, lade_distributed, utils  # noqa: F401

sys.modules[__name__ + ".decoding"] = decoding
sys.modules[__name__ + ".utils"] = utils
sys.modules[__name__ + ".lade_distributed"] = lade_distributed
__all__ = _impl.__all__
