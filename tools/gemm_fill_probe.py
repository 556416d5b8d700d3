"""Is the 128-row class bound per CU or chip-wide?  The unsplit gate/up GEMM (SwiGLU epilogue) of the 13B width at 120 rows on synthetic
N: the same 128 weight rows per work-group on 216 work-groups (the real N = 27648: 40 CUs idle) and on 256 (N = 32768: 18.5 % more
bytes); 96 rows per work-group on 256 and on 216 work-groups.  Per-CU bound: time follows the rows per work-group, not the bytes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import ops

M, K = int(os.environ.get("M", "120")), int(os.environ.get("K", "5120"))
mb = (M + 31) // 32


def timeit(fn, reps=40, rounds=5):
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


a = torch.randn(M, K, device="cuda").bfloat16()
for rep in range(2):
    cases = ((128, 2, 1, 216), (128, 2, 1, 256), (128, 2, 1, 240), (96, 2, 1, 256), (96, 2, 1, 216), (64, 2, 1, 256))
    if os.environ.get("CASES"):            # "bn:mt:nt:wgs,..."
        cases = tuple(tuple(int(x) for x in c.split(":")) for c in os.environ["CASES"].split(","))
    for (bn, mt, nt, wgs) in cases:
        N = bn * wgs
        n_w = max(3, int(700e6 / (N * K * 2)))
        kts = [ops.to_ktile((torch.randn(N, K, device="cuda") * 0.02).bfloat16()) for _ in range(n_w)]
        act = torch.empty(M, N // 2, dtype=torch.bfloat16, device="cuda")
        i = [0]

        def run():
            i[0] = (i[0] + 1) % n_w
            ops.gemm_swiglu(a, kts[i[0]], act, bn, mb, mt, nt, int(os.environ.get('RING', '3')))
        t = timeit(run)
        print(f"M={M} K={K} bn={bn} x {wgs} work-groups (N={N}, {N * K * 2 / 1e6:.0f} MB): {t:6.2f} us  {N * K * 2 / 1e6 / t:4.2f} TB/s  {bn * K * 2 / 1e3 / t:5.1f} GB/s per work-group", flush=True)
        del kts
        torch.cuda.empty_cache()
