"""What the activation re-reads cost the weight-streaming GEMM: the same launches with one operand's traffic removed
(LADE_GEMM_DBG bit 64: every activation piece re-reads one line; bit 128: every weight piece does), with and without the MFMA /
store phases (bits 4 / 1).  Run once per LADE_GEMM_DBG value: 0, 64, 128, 5, 69, 133."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd.cabi import call, dtype_code, ptr

M = int(os.environ.get("M", "60"))


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
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


line = []
# (name, N, K, S, bn, mb, mt, nt, epilogue): the 7B step's choices at 60 rows
for name, N, K, S, bn, mb, mt, nt, epi in (("qkv", 12288, 4096, 4, 192, 2, 2, 0, 0), ("o", 4096, 4096, 8, 128, 2, 1, 0, 0),
                                           ("gate_up(fused)", 22016, 4096, 1, 96, 2, 1, 1, 1), ("down", 4096, 11008, 8, 128, 2, 1, 0, 0)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    ws = [torch.randn(N, K, device="cuda").bfloat16() * 0.02 for _ in range(max(2, int(600e6 / (N * K * 2))))]
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    part = torch.empty(max(S, 2), M, N, dtype=torch.float32, device="cuda")
    i = [0]

    def mine():
        i[0] = (i[0] + 1) % len(ws)
        call("lade_gemm_skinny", ptr(a), a.stride(0), ptr(ws[i[0]]), ws[i[0]].stride(0), ptr(out) if S == 1 else None, out.stride(0) if S == 1 else 0,
             ptr(part), M, N, K, S, bn, mb, mt, nt, 0, epi, dtype_code(a))

    t = timeit(mine)
    wg = -(-N // bn) * S
    a_mb = wg * 64 * (K / S) * 2 / 1e6
    line.append(f"{name} {t:6.2f} us (W {N * K * 2 / 1e6:.0f} MB, A re-reads {a_mb:.0f} MB, {wg} WGs)")
print(f"LADE_GEMM_DBG={os.environ.get('LADE_GEMM_DBG', '0'):>3s} M={M}: " + " | ".join(line), flush=True)
