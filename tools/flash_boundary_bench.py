"""What the operator-level drop-in costs (run on the GPU box): `flash_attn_func(q, k, v, ..., lookahead=[7 ints])` of
lookaheaddecoding_amd/flash_attn_lade.py at the BASELINE shapes - the K / V re-layout (lade_kv_pack_bshd: the whole K / V per call, as the
reference's own per-layer torch.cat re-copies it) against the attention launch pair it feeds.  K / V rotate over enough buffers to come from HBM."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import cabi
from lookaheaddecoding_amd.flash_attn_lade import flash_attn_func, lookahead_tuple, _workspace


def timed(fn, reps):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (name, H, Hkv, W, N, g, P) in (("c2 (7B heads, W=15 N=5, cold)", 32, 32, 15, 5, 0, 2219), ("c2 with candidates", 32, 32, 15, 5, 15, 2219),
                                   ("c4 (13B heads, W=20 N=7)", 40, 40, 20, 7, 0, 2219), ("c5 (70B heads, GQA 8)", 64, 8, 15, 5, 0, 2219)):
    d, gs = 128, N - 1
    ls = [W - 1] + [W] * (N - 2)
    T = 1 + sum(ls) + g * gs
    S = P + T
    n_buf = max(2, min(64, 600_000_000 // (2 * S * Hkv * d * 2)))
    q = torch.randn(1, T, H, d, device="cuda").bfloat16()
    # as the reference hands them over: transposed VIEWS of [1, Hkv, S, d] tensors (modeling_llama.py:636-638)
    ks = [torch.randn(1, Hkv, S, d, device="cuda").bfloat16().transpose(1, 2) for _ in range(n_buf)]
    vs = [torch.randn(1, Hkv, S, d, device="cuda").bfloat16().transpose(1, 2) for _ in range(n_buf)]
    tup = lookahead_tuple(1, ls, g, P)
    us_all = timed(lambda i: flash_attn_func(q, ks[i % n_buf], vs[i % n_buf], 0.0, causal=True, lookahead=tup), 200)
    kc, vt = _workspace(q.device, q.dtype, Hkv, d, S)

    def pack(i):
        kk, vv = ks[i % n_buf][0], vs[i % n_buf][0]
        cabi.call("lade_kv_pack_bshd", cabi.ptr(kk), cabi.ptr(vv), kk.stride(0), kk.stride(1), cabi.ptr(kc), cabi.ptr(vt), S, Hkv, d, kc.shape[1], 2)

    us_pack = timed(pack, 200)
    mb = 2 * S * Hkv * d * 2 / 1e6
    print(f"{name:32s} T={T:4d} S={S:5d}: flash_attn_func {us_all:7.1f} us per call = re-layout {us_pack:6.1f} us ({mb:5.1f} MB read + {mb:5.1f} MB written: "
          f"{2 * mb / us_pack:5.2f} TB/s) + attention pair {us_all - us_pack:6.1f} us", flush=True)
