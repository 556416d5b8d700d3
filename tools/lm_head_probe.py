"""lm_head of a decode step (rows that are read only: 1 for plain decoding, 16 ... 76 for a lookahead step) on the library GEMM against the
skinny GEMM writing the model dtype directly (n_split = 1: N / bn work-groups), on row-major and on K-tile-major weights.  Two copies of the
weight are used in alternation (262 MB each: more than the 256 MB Infinity Cache).   python tools/lm_head_probe.py [V] [K]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import ops

V = int(sys.argv[1]) if len(sys.argv) > 1 else 32000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096


def timeit(fn, reps=20, rounds=4):
    for _ in range(2):
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


rows = [(torch.randn(V, K, device="cuda") * 0.02).bfloat16() for _ in range(3)]
kts = [ops.to_ktile(w) for w in rows]
mbytes = V * K * 2 / 1e6
for M in (1, 16, 31, 46, 76):
    a = torch.randn(M, K, device="cuda").bfloat16()
    out = torch.empty(M, V, dtype=torch.bfloat16, device="cuda")
    i = [0]

    def lib():
        i[0] = (i[0] + 1) % len(rows)
        torch.matmul(a, rows[i[0]].t(), out=out)

    t_lib = timeit(lib)
    ref = out.clone()
    mb = 1 if M <= 32 else 2 if M <= 64 else 3
    res = []
    for bn in (64, 96, 128, 192, 224, 256):
        for mt in sorted({1, mb}):
            for nt in (0, 1, 2):
                for lay, ws in (("row", rows), ("kt", kts)):
                    def mine():
                        i[0] = (i[0] + 1) % len(ws)
                        ops.gemm_skinny(a, ws[i[0]], out=out, n_split=1, bn=bn, mb=mb, mt=mt, nt=nt)
                    try:
                        t = timeit(mine)
                    except Exception:
                        continue
                    res.append((t, lay, bn, mt, nt))
    i[0] = len(rows) - 1
    best_kt = min(r for r in res if r[1] == "kt")
    best_row = min(r for r in res if r[1] == "row")
    ops.gemm_skinny(a, kts[(i[0] + 1) % len(kts)], out=out, n_split=1, bn=best_kt[2], mb=mb, mt=best_kt[3], nt=best_kt[4])
    i[0] = len(rows) - 1
    lib_out = torch.matmul(a, rows[0].t())
    err = (out.float() - lib_out.float()).abs().max().item()
    print(f"V={V} K={K} rows={M:3d}: library {t_lib:6.2f} us {mbytes / t_lib:5.2f} TB/s | skinny row-major {best_row[0]:6.2f} us {best_row[2:]} | "
          f"skinny K-tile-major {best_kt[0]:6.2f} us {mbytes / best_kt[0]:5.2f} TB/s {best_kt[2:]}  ({(best_kt[0] / t_lib - 1) * 100:+5.1f} % vs library; max |diff| {err:.3g})", flush=True)
    for r in sorted(r for r in res if r[1] == "kt")[:4]:
        print(f"      kt bn={r[2]} mt={r[3]} nt={r[4]}: {r[0]:6.2f} us", flush=True)
