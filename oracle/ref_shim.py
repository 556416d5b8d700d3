"""Import shim that loads the UNMODIFIED reference (`/root/reference/lade`) in this container.

TEST INFRASTRUCTURE ONLY.  Used by `oracle/make_golden.py` (and nothing else) to run the
reference's own code on CPU and freeze its outputs as fixtures under `tests/golden/`.
`/root/reference` does not exist on the GPU box, so nothing under `tests/`, `bench.py` or
`__graft_entry__.py` may import this module at run time.

Why a shim is needed: the reference pins transformers 4.36.2; this image has
5.15.  `lade/decoding.py:8` imports `GreedySearchOutput`/`SampleOutput`, and
`lade/models/modeling_llama.py:50` imports `is_torch_fx_available`; both are gone in 5.x.
The shim only adds placeholder attributes to *transformers* before the import - the
reference's files are not edited and not copied.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def load_reference():
    """Returns the reference `lade` package (its own code, executed from /root/reference)."""
    import transformers
    import transformers.generation.utils as gu
    import transformers.utils.import_utils as iu
    import transformers.utils as tu

    for mod in (iu, tu):
        if not hasattr(mod, "is_torch_fx_available"):
            mod.is_torch_fx_available = lambda: False
    for name in ("GreedySearchOutput", "SampleOutput"):
        if not hasattr(gu, name):
            setattr(gu, name, type(name, (), {}))
    from transformers import GenerationMixin
    for name in ("greedy_search", "sample"):
        if not hasattr(GenerationMixin, name):
            setattr(GenerationMixin, name, lambda self, *a, **k: (_ for _ in ()).throw(
                NotImplementedError("placeholder for HF<=4.36 GenerationMixin." + name)))
    # our own drop-in package is also importable as `lade`; make sure the reference wins here
    for k in [k for k in sys.modules if k == "lade" or k.startswith("lade.")]:
        del sys.modules[k]
    if REFERENCE_ROOT in sys.path:
        sys.path.remove(REFERENCE_ROOT)
    sys.path.insert(0, REFERENCE_ROOT)
    import lade  # noqa: E402  (the reference)
    assert lade.__file__.startswith(REFERENCE_ROOT), lade.__file__
    return lade
