"""CPU: the C-ABI library builds, loads and exports every symbol include/lade_hip.h declares;
argument validation works without a GPU (no compute calls)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "lade_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lade_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from lookaheaddecoding_amd import cabi
    if not os.path.exists(cabi.LIB_PATH):
        g.build()
    return cabi.load_library()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from lookaheaddecoding_amd import cabi
    syms = declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in lade_hip.h but not exported"
        assert s in cabi.SIGNATURES, f"{s} has no ctypes signature"
    assert sorted(cabi.SIGNATURES) == syms
    assert lib.lade_version() == cabi.ABI_VERSION == 3


def test_argument_validation_returns_error_codes_not_crashes(lib):
    from lookaheaddecoding_amd import cabi
    a = cabi.AttnArgs()
    rc = lib.lade_attn_fwd(C.byref(a), None)
    assert rc == -1 and b"null" in lib.lade_last_error_string()
    rc = lib.lade_argmax_rows(None, 10, 1, 10, 0, None, None)
    assert rc == -1
    rc = lib.lade_pool_lookup(None, None, 10, 0, 3, None, None, None, None)
    assert rc == -1 and b"lade_pool_lookup" in lib.lade_last_error_string()


def test_missing_library_fails_loudly(tmp_path):
    from lookaheaddecoding_amd import cabi
    old = cabi._lib
    cabi._lib = None
    try:
        with pytest.raises(cabi.LadeHipError):
            cabi.load_library(str(tmp_path / "nope.so"))
    finally:
        cabi._lib = old


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lookaheaddecoding_amd import cabi
    from lookaheaddecoding_amd.engine import StepEngine
    from lookaheaddecoding_amd.weights import make_config, random_weights_numpy
    cfg = make_config("tiny-d64")
    w = {k: torch.as_tensor(v) for k, v in random_weights_numpy(cfg).items()}
    with pytest.raises(cabi.LadeHipError):
        StepEngine(cfg, w, dtype=torch.float32)
