"""Would a qkv GEMM WITHOUT split-K (so that RoPE + the KV append could live in its epilogue, as SwiGLU does in the gate/up GEMM) stream fast
enough?  Times lade_gemm_skinny on the 7B / 13B qkv shapes at 60 / 120 rows with 64- / 96- / 128-row weight blocks, unsplit and split
(run on the GPU box: python tools/gemm_qkv_unsplit_probe.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd.cabi import LadeHipError, call, dtype_code, ptr


def timeit(fn, reps=40, rounds=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


for name, N, K, M in (("qkv-7B", 12288, 4096, 60), ("qkv-13B", 15360, 5120, 120)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    ws = [torch.randn(N, K, device="cuda").bfloat16() * 0.02 for _ in range(max(2, int(600e6 / (N * K * 2))))]
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    part = torch.empty(8, M, N, dtype=torch.float32, device="cuda")
    i = [0]
    mb = (M + 31) // 32
    for (S, bn, mt, nt) in ((1, 64, 1, 1), (1, 64, 2, 1), (1, 96, 1, 1), (1, 96, 2, 1), (1, 128, 1, 0), (1, 128, 2, 0), (2, 64, 1, 1), (2, 64, 2, 1), (2, 96, 2, 1), (3, 64, 1, 1), (4, 128, 2, 0), (5, 64, 1, 1), (5, 128, 2, 0), (6, 64, 1, 1)):
        def mine():
            i[0] = (i[0] + 1) % len(ws)
            call("lade_gemm_skinny", ptr(a), a.stride(0), ptr(ws[i[0]]), ws[i[0]].stride(0), ptr(out) if S == 1 else None, out.stride(0) if S == 1 else 0,
                 ptr(part) if S > 1 else None, M, N, K, S, bn, mb, mt, nt, 0, 0, dtype_code(a))
        try:
            t = timeit(mine)
            print(f"{name} M={M} S={S} bn={bn} mb={mb} mt={mt} nt={nt}: {t:6.2f} us {N * K * 2 / t / 1e6:5.2f} TB/s  ({N // bn * S} work-groups)", flush=True)
        except LadeHipError as e:
            print(name, S, bn, mb, mt, nt, "not built:", str(e)[:70])
