"""Skinny GEMM microbench vs torch.matmul (hipBLASLt): python tools/gemm_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import ops


NO_REDUCE = bool(int(os.environ.get('NO_REDUCE', '0')))


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


shapes = [("qkv", 12288, 4096), ("o", 4096, 4096), ("gate_up", 22016, 4096), ("down", 4096, 11008), ("lm_head", 32000, 4096)]
for M in [int(x) for x in os.environ.get('M', '60,120').split(',')]:
    for name, N, K in shapes:
        a = torch.randn(M, K, device="cuda").bfloat16()
        # a few distinct weight copies so the stream comes from HBM, not from the 256 MB Infinity Cache
        ws = [torch.randn(N, K, device="cuda").bfloat16() * 0.02 for _ in range(max(1, int(600e6 / (N * K * 2))))]
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        i = [0]

        def ref():
            i[0] = (i[0] + 1) % len(ws)
            torch.matmul(a, ws[i[0]].t(), out=out)

        t_ref = timeit(ref)
        res = []
        want = torch.matmul(a.float(), ws[0].float().t())
        # (mb, mt, bn, nt): nt = 0 -> as many n-groups as waves allow
        shapes_ = ([(2, 1, 64, 0), (2, 1, 128, 0), (2, 1, 256, 0), (2, 2, 128, 0), (2, 2, 192, 0), (2, 2, 256, 0), (2, 2, 128, 2), (2, 2, 192, 2), (2, 2, 256, 2),
                    (2, 2, 192, 3), (2, 2, 256, 4)] + ([(1, 1, 64, 0), (1, 1, 128, 0), (1, 1, 256, 0), (1, 1, 128, 2), (1, 1, 256, 2)] if M <= 32 else [])
                   if M <= 64 else [(4, 1, 128, 0), (4, 1, 192, 0), (4, 1, 256, 0), (4, 2, 128, 0), (4, 2, 192, 0), (4, 2, 256, 0), (4, 2, 128, 2), (4, 4, 128, 0),
                                    (4, 4, 192, 0), (4, 4, 256, 0), (4, 4, 192, 2), (4, 4, 256, 2), (4, 4, 128, 2)]
                   + ([(3, 1, 128, 0), (3, 1, 192, 0), (3, 3, 128, 0), (3, 3, 192, 0), (3, 3, 256, 0), (3, 3, 192, 2), (3, 3, 256, 2)] if M <= 96 else []))
        for mb, MT, bn, NTW in shapes_:
            for S in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
                part = torch.empty(S, M, N, dtype=torch.float32, device="cuda") if S > 1 else None

                def mine():
                    i[0] = (i[0] + 1) % len(ws)
                    if NO_REDUCE and S > 1:
                        from lookaheaddecoding_amd.cabi import call, ptr, dtype_code
                        call('lade_gemm_skinny', ptr(a), a.stride(0), ptr(ws[i[0]]), ws[i[0]].stride(0), ptr(out), out.stride(0), ptr(part), M, N, K, S, bn, mb, MT, NTW, 0, 0, dtype_code(a))
                    else:
                        ops.gemm_skinny(a, ws[i[0]], out=out, n_split=S, bn=bn, part=part, mb=mb, mt=MT, nt=NTW)

                try:
                    got = ops.gemm_skinny(a, ws[0], n_split=S, bn=bn, mb=mb, mt=MT, nt=NTW).float()
                    err = (got - want).abs().max().item()
                    t = timeit(mine)
                    res.append((t, f"{mb}.{MT}x{bn}.{NTW}", S, err))
                except Exception as ex:
                    res.append((float("inf"), f"{mb}.{MT}x{bn}.{NTW}", S, 0.0))
        res.sort(key=lambda x: x[0])
        assert max(r[3] for r in res) < 0.2, max(res, key=lambda r: r[3])
        mb = N * K * 2 / 1e6
        best = res[0]
        print(f"M={M:3d} {name:8s} N={N:5d} K={K:5d}  W={mb:6.1f} MB  torch {t_ref:7.2f} us ({mb / t_ref:5.2f} TB/s) | best mine {best[0]:7.2f} us "
              f"({mb / best[0]:5.2f} TB/s) bn={best[1]} S={best[2]} err={best[3]:.3g} | " + " ".join(f"[{b}/{s}:{t:.1f}]" for t, b, s, _ in res[1:6]), flush=True)
