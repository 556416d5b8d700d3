"""Steps / prefill chunks WIDER than 256 rows on the hand-written GEMM (several 256-row blocks per launch, K-tile-major weights, no split-K)
against the library GEMM (hipBLASLt through torch.matmul on the row-major weight), per projection of a 7B / 13B / 70B layer.  Every launch
on a different layer's weights; 12 dependent launches per hipGraph.  Checks each result against the fp32 product.
    python tools/gemm_wide_probe.py [7b|13b|70b] [M ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import cabi, ops

MODEL = sys.argv[1] if len(sys.argv) > 1 else "7b"
MS = [int(x) for x in sys.argv[2:]] or [512, 847, 2304]
HID, INTER, QKV = {"7b": (4096, 11008, 12288), "13b": (5120, 13824, 15360), "70b": (8192, 28672, 10240)}[MODEL]
DT = torch.float16 if os.environ.get("PROBE_DTYPE") == "f16" else torch.bfloat16
# (bn, mb, mt, nt): wave grids built for 256-row blocks (gemm_kernel.hpp)
CANDS = [(128, 8, 4, 1, 0), (128, 8, 4, 2, 0), (128, 8, 2, 2, 0), (64, 8, 4, 1, 0), (256, 8, 4, 2, 0), (256, 8, 2, 4, 0)]
if cabi.experimental():          # the ping-pong K loop (csrc/gemm_pp.hpp, ring = 10 + stages) at its 256-row shapes: the compute-bound regime it was NOT measured in
    CANDS += [(128, 8, 4, 2, 10), (128, 8, 4, 2, 12), (128, 8, 4, 2, 14), (64, 8, 4, 1, 10), (64, 8, 4, 1, 14), (64, 8, 4, 1, 16)]


def timeit(fn, reps=12, rounds=3):
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


for M in MS:
    tot_lib, tot_best = 0.0, 0.0
    for name, N, K, swiglu in (("qkv", QKV, HID, False), ("o", HID, HID, False), ("gate_up", 2 * INTER, HID, True), ("down", HID, INTER, False)):
        n_w = max(2, min(4, int(600e6 / (N * K * 2))))
        rows = [(torch.randn(N, K, device="cuda") * 0.02).to(DT) for _ in range(n_w)]
        if swiglu:                                  # the engine's fused gate/up order (16-row interleave): the SwiGLU epilogue is lane-local
            rows = [ops.interleave_gate_up(w[:N // 2].contiguous(), w[N // 2:].contiguous()) for w in rows]
        kts = [ops.to_ktile(w) for w in rows]
        a = torch.randn(M, K, device="cuda").to(DT)
        out = torch.empty(M, N, dtype=DT, device="cuda")
        act = torch.empty(M, N // 2, dtype=DT, device="cuda")
        i = [0]

        def lib():
            i[0] = (i[0] + 1) % n_w
            torch.matmul(a, rows[i[0]].t(), out=out)
            if swiglu:
                ops.silu_mul(out, out=act, layout=1)

        t_lib = timeit(lib)
        i[0] = -1
        lib()
        ref_lib = (act if swiglu else out).clone()
        # fp32 reference of weight 0 on a sample of rows
        rs = torch.arange(0, M, max(1, M // 16), device="cuda")
        ref32 = a[rs].float() @ rows[0].float().t()
        if swiglu:
            ref32 = ops.silu_mul(ref32.to(DT), layout=1).float()
        res = []
        for (bn, mb, mt, nt, ring) in CANDS:
            def run():
                i[0] = (i[0] + 1) % n_w
                if swiglu:
                    ops.gemm_swiglu(a, kts[i[0]], act, bn, mb, mt, nt, ring)
                else:
                    ops.gemm_skinny(a, kts[i[0]], out=out, n_split=1, bn=bn, mb=mb, mt=mt, nt=nt, ring=ring)
            try:
                t = timeit(run)
            except cabi.LadeHipError as e:
                continue
            i[0] = -1
            run()
            got = (act if swiglu else out)
            err = (got[rs].float() - ref32).abs().max().item()
            err_lib = (ref_lib[rs].float() - ref32).abs().max().item()
            res.append((t, (bn, mb, mt, nt, ring), err, err_lib))
        flops = 2.0 * M * N * K
        res.sort()
        tot_lib += t_lib
        tot_best += min(t_lib, res[0][0]) if res else t_lib
        line = " ".join(f"{c}:{t:.1f}us/{flops / t / 1e6:.0f}TF(err {e:.3g} lib {el:.3g})" for t, c, e, el in res[:4])
        print(f"{MODEL} M={M} {name:8s} N={N} K={K}: library {t_lib:7.1f} us {flops / t_lib / 1e6:5.0f} TF | {line}", flush=True)
        del rows, kts
        torch.cuda.empty_cache()
    print(f"{MODEL} M={M} layer sum: library {tot_lib:7.1f} us | best of both {tot_best:7.1f} us", flush=True)
