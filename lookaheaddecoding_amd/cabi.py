"""ctypes binding of liblade_hip.so (the C ABI declared in include/lade_hip.h).

The library is the product's compute path.  There is NO fallback: if the shared object is missing
or a call fails, a `LadeHipError` is raised.  Tensors are passed as raw device pointers
(`tensor.data_ptr()`) together with torch's current HIP stream, so launches are ordered with the
surrounding torch ops (GEMMs) and are capturable in a torch.cuda.CUDAGraph (= hipGraph).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LADE_HIP_LIB") or os.path.join(_HERE, "liblade_hip.so")     # env override: kernel experiments

ABI_VERSION = 3           # LADE_ABI_VERSION of include/lade_hip.h this binding was written against (checked at load time)
LADE_BF16, LADE_F16, LADE_F32 = 0, 1, 2
DTYPE_CODE = {torch.bfloat16: LADE_BF16, torch.float16: LADE_F16, torch.float32: LADE_F32}

# control-block indices (include/lade_hip.h)
CTL_P, CTL_LST_TOKEN, CTL_LST_POS, CTL_N_INPUT, CTL_G, CTL_FILL_LEVEL = 0, 1, 2, 3, 4, 5
CTL_MAX_HIT, CTL_MAX_HIT_IDX, CTL_N_ACCEPT, CTL_STEP = 6, 7, 8, 9
CTL_KV_SRC, CTL_KV_DST, CTL_KV_CNT, CTL_FIRST_GUESS = 10, 11, 12, 13
CTL_HITS, CTL_WLEN, CTL_WORDS = 16, 32, 64
MAX_LEVEL, MAX_WINDOW, MAX_GUESS_SET = 16, 128, 64
REC_WORDS = 8 + MAX_LEVEL


def record_seal(rec, step_no: int) -> int:
    """lade_record_seal over a snapshot of a step record (list / array of REC_WORDS int32): what its last word must be"""
    arr = (C.c_uint32 * REC_WORDS)(*[int(x) & 0xFFFFFFFF for x in rec])
    return int(lib().lade_record_seal(arr, int(step_no) & 0xFFFFFFFF))


class LadeHipError(RuntimeError):
    pass


class MaskParams(C.Structure):
    _fields_ = [("T", C.c_int32), ("P", C.c_int32), ("is_prefill", C.c_int32), ("s", C.c_int32), ("lguess", C.c_int32),
                ("gs", C.c_int32), ("level_offset", C.c_int32), ("dist_offset", C.c_int32), ("layout", C.c_int32)]


class AttnArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k_cache", C.c_void_p), ("vt_cache", C.c_void_p), ("out", C.c_void_p),
                ("part_o", C.c_void_p), ("part_ml", C.c_void_p), ("dyn_P", C.c_void_p),
                ("q_row_stride", C.c_int64), ("out_row_stride", C.c_int64),
                ("H", C.c_int32), ("Hkv", C.c_int32), ("d", C.c_int32), ("S_max", C.c_int32),
                ("dtype", C.c_int32), ("n_splits", C.c_int32), ("scale", C.c_float), ("mask", MaskParams),
                # ABI 2: work-group shape + fused RoPE / KV append (all zero = the ABI-1 behaviour)
                ("wg_rows", C.c_int32), ("n_parts", C.c_int32), ("qkv_parts", C.c_void_p), ("part_stride", C.c_int64),
                ("positions", C.c_void_p), ("cos_tab", C.c_void_p), ("sin_tab", C.c_void_p), ("max_pos", C.c_int32),
                ("sync_flags", C.c_void_p)]


_lib: Optional[C.CDLL] = None

_i32, _i64, _f32, _vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p

# name -> argtypes; every symbol of include/lade_hip.h (tests/test_cabi_symbols.py checks the list)
SIGNATURES = {
    "lade_attn_fwd": [C.POINTER(AttnArgs), _vp],
    "lade_attn_combine": [C.POINTER(AttnArgs), _vp],
    "lade_mask_render": [C.POINTER(MaskParams), _vp, _vp],
    "lade_rope_kv_append": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "lade_rope_rows_dynamic": [_vp, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _vp],
    "lade_kv_commit": [_vp, _i64, _i64, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _vp],
    "lade_kv_pack_bshd": [_vp, _vp, _i64, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "lade_build_inputs": [_vp, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _vp],
    "lade_argmax_rows": [_vp, _i64, _i32, _i32, _i32, _vp, _vp],
    "lade_argmax_pairs": [_vp, _i32, _i32, _vp, _vp],
    "lade_verify_greedy": [_vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp],
    "lade_pool_insert_window": [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _i32, _i32, _vp],
    "lade_pool_insert_ngrams": [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp],
    "lade_pool_fill_prompt": [_vp, _vp, _i32, _i32, _i32, _vp, _i32, _vp],
    "lade_pool_lookup": [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "lade_window_fill_first": [_vp, _i32, _vp, _vp, _i32, _vp],
    "lade_window_fill": [_vp, _i32, _vp, _i32, _vp, _i32, _vp],
    "lade_window_roll": [_vp, _i32, _vp, _vp, _i32, _i32, _vp],
    "lade_greedy_post_step": [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _vp,
                              _i32, _vp, _vp, _vp, _vp, _vp],
    "lade_lp_pack": [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _vp],
    "lade_lp_reduce_apply": [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _i32, _vp],
    "lade_lp_unique_id": [_vp],
    "lade_lp_comm_create": [_vp, _i32, _i32, C.POINTER(_vp)],
    "lade_lp_allgather": [_vp, _vp, _vp, _i32, _vp],
    "lade_lp_comm_count": [_vp, C.POINTER(_i32)],
    "lade_lp_comm_destroy": [_vp],
    "lade_softmax_rows": [_vp, _i64, _i32, _i32, _i32, _f32, _vp, _vp],
    "lade_softmax_gather": [_vp, _i64, _i32, _i32, _i32, _f32, _i32, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "lade_warp_rows": [_vp, _i64, _i32, _i32, _i32, _f32, _i32, _f32, _i32, _vp, _vp],
    "lade_rmsnorm": [_vp, _vp, _vp, _i32, _i32, _f32, _i32, _vp],
    "lade_embed_rmsnorm": [_vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _f32, _i32, _vp],
    "lade_add_rmsnorm": [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _i32, _vp],
    "lade_add_rmsnorm_rows": [_vp, _vp, _vp, _i32, _i64, _vp, _vp, _vp, _i32, _i32, _i32, _f32, _i32, _vp],
    "lade_silu_mul": [_vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "lade_gather_rows": [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp],
    "lade_gemm_skinny": [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "lade_gemm_skinny_kt": [_vp, _i64, _vp, _vp, _i64, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "lade_gemm_ra_kt": [_vp, _i64, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "lade_weight_to_ktile": [_vp, _i64, _vp, _i32, _i32, _i32, _vp],
    "lade_weight_from_ktile": [_vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "lade_add_rmsnorm_parts": [_vp, _vp, _i32, _i64, _vp, _vp, _i32, _i32, _f32, _i32, _vp],
    "lade_silu_mul_parts": [_vp, _i32, _i64, _vp, _i32, _i32, _i32, _i32, _vp],
    "lade_rope_kv_append_parts": [_vp, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "lade_splitk_reduce": [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp],
    "lade_record_seal": [C.POINTER(C.c_uint32), C.c_uint32],
    "lade_version": [],
    "lade_build_flags": [],
    "lade_last_error_string": [],
    "lade_time_attn": [C.POINTER(AttnArgs), _i32, C.POINTER(C.c_float), _vp],
    "lade_time_attn_rot": [C.POINTER(AttnArgs), _i32, _i32, C.POINTER(C.c_float), _vp],
}


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """Loads liblade_hip.so and binds every exported symbol.  Raises LadeHipError when absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise LadeHipError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"or `make -C lookaheaddecoding_amd/csrc` (the HIP extension is required; there is no CPU fallback)")
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise LadeHipError(f"cannot load {path}: {e}") from e
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise LadeHipError(f"{path} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = C.c_char_p if name == "lade_last_error_string" else (C.c_uint32 if name == "lade_record_seal" else C.c_int)
    got = lib.lade_version()
    if got != ABI_VERSION:        # argument lists changed between versions: a stale build would read `epilogue` as `ring`, and so on
        raise LadeHipError(f"{path} implements ABI version {got}, this binding expects {ABI_VERSION}: rebuild it (make -C lookaheaddecoding_amd/csrc)")
    _lib = lib
    return lib


def lib() -> C.CDLL:
    return load_library()


def debug(name: str, default=None):
    """One experiment switch of `LADE_DEBUG=name[=value],name2[=value2],...` (read at every call: tests flip them at run time): the kernel
    ablation bits and tuner experiments that only tools/ and A/B runs use - gemm_dbg, attn_dbg, attn_shape, attn_splits, attn_split_rule,
    combine_hpb (those six are read by the library itself), tune_consumer, gu_tail_fixed, tune_ring, tune_verbose, attn_tune, row_classes,
    draw_torch, fuse_tail, fuse_rope, poll_reads_per_yield.  A bare name means "1"."""
    s = os.environ.get("LADE_DEBUG")
    if not s:
        return default
    for item in s.split(","):
        k, _, v = item.strip().partition("=")
        if k == name:
            return v if v != "" else "1"
    return default


def experimental() -> bool:
    """whether the loaded library was built with -DLADE_EXPERIMENTAL (the attention forms with RoPE + KV append inside the launch)"""
    return bool(lib().lade_build_flags() & 1)


def stream_ptr() -> int:
    """Raw hipStream_t of torch's current stream on the current device (the C-level getter: an eager step issues
    several hundred entry-point calls, torch.cuda.current_stream() would cost more than the launches themselves)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().lade_last_error_string()
        raise LadeHipError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def dtype_code(t: torch.Tensor) -> int:
    try:
        return DTYPE_CODE[t.dtype]
    except KeyError:
        raise LadeHipError(f"unsupported dtype {t.dtype}")


def call_plain(name: str, *args) -> None:
    """Invokes an entry point that takes no stream (communicator management) and raises on failure."""
    fn = getattr(lib(), name)
    check(fn(*args), name)


def call(name: str, *args) -> None:
    """Invokes an entry point with the current torch stream appended and raises on failure."""
    fn = getattr(lib(), name)
    check(fn(*args, stream_ptr()), name)
