"""lade/lade_distributed.py:5-12"""
from .decoding import CONFIG_MAP


def get_device():
    if "LOCAL_RANK" not in CONFIG_MAP:
        return 0
    return CONFIG_MAP["LOCAL_RANK"]


def distributed():
    return "DIST_WORKERS" in CONFIG_MAP and CONFIG_MAP["DIST_WORKERS"] > 1
