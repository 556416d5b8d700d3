"""The 16-row-granular skinny GEMM (csrc/gemm16.hpp: v_mfma_f32_16x16x32 tiles, 80 / 112 / 144 / 176-row activation tiles) against the 32-row classes a
step of that many rows pads to today (96 / 128 / 160 / 192), isolated: split-K projections of a 7B / 13B layer at M rows, K-tile-major weights,
hipGraph of 40 launches over rotating weights; every 16-row configuration CHECKED against an fp64 product.
    MODEL=7b M=76 python tools/gemm16_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import cabi, ops

M = int(os.environ.get("M", "76"))
MODEL = os.environ.get("MODEL", "7b")
DT = {"bf16": torch.bfloat16, "f16": torch.float16}[os.environ.get("DTYPE", "bf16")]
HID, INTER, QKV = {"7b": (4096, 11008, 12288), "13b": (5120, 13824, 15360), "70b": (8192, 28672, 10240)}[MODEL]
mb16 = (M + 15) // 16
mb32 = (M + 31) // 32
if mb32 == 7:
    mb32 = 8


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


LOCK = {3: ((3, 1, 0, (64, 128, 192)), (3, 3, 0, (128, 192, 256)), (3, 3, 2, (192, 256)), (3, 3, 1, (96,))),
        4: ((4, 1, 0, (64, 128, 192)), (4, 2, 0, (128, 192, 256)), (4, 4, 0, (192, 256)), (4, 4, 2, (192, 256)), (4, 2, 1, (96,))),
        5: ((5, 5, 1, (96, 128, 192, 256)), (5, 5, 2, (128, 256))),
        6: ((6, 3, 1, (64, 128)), (6, 3, 2, (128,)), (6, 2, 1, (64,)), (6, 2, 2, (128,))),
        2: ((2, 1, 0, (128, 256)), (2, 2, 0, (128, 192, 256)), (2, 1, 1, (64, 96)))}[mb32]
print(f"{MODEL} M={M} {DT}: {16 * mb16}-row tile of 16x16x32 MFMAs vs the {32 * mb32}-row class; us per launch, TB/s of weights", flush=True)
for name, N, K in (("qkv", QKV, HID), ("o", HID, HID), ("gate_up(split)", 2 * INTER, HID), ("down", HID, INTER)):
    a = torch.randn(M, K, device="cuda").to(DT)
    n_w = max(3, int(700e6 / (N * K * 2)))
    kts = [ops.to_ktile((torch.randn(N, K, device="cuda") * 0.02).to(DT)) for _ in range(n_w)]
    ref = a.double() @ ops.from_ktile(kts[0]).double().t()
    part = torch.empty(16 * 128 * N, dtype=torch.float32, device="cuda")
    wbytes = N * K * 2
    i = [0]

    def rot():
        i[0] = (i[0] + 1) % n_w
        return kts[i[0]]

    def splits(bn):
        nblk = (N + bn - 1) // bn
        return sorted({S for S in {max(1, round(256 / nblk)), max(1, round(384 / nblk)), max(1, round(512 / nblk)), max(1, round(768 / nblk))}
                       if 2 <= S <= 16 and K // 64 >= 2 * S and S * M * N <= part.numel()})

    rows = []
    for (mb, mt, nt, bns) in LOCK:
        for bn in bns:
            for S in splits(bn):
                for ring in (0, 2):
                    try:
                        t = timeit(lambda: ops.gemm_parts(a, rot(), part, S, bn, mb, mt, nt, ring))
                    except cabi.LadeHipError:
                        continue
                    rows.append((t, f"32-row blocks ({32 * mb} rows)  S={S} bn={bn} mt={mt} nt={nt} ring={ring}", None))
    for nt16, bn in ((1, 128), (2, 256), (2, 128), (3, 384), (4, 256)):
        if N % bn and bn > 256:
            continue
        for S in splits(bn):
            for ring in (0, 2, 3):
                try:
                    t = timeit(lambda: ops.gemm_parts(a, rot(), part, S, bn, mb16, 16, nt16, ring))
                except cabi.LadeHipError:
                    continue
                rows.append((t, f"16-ROW TILES ({16 * mb16} rows)  S={S} bn={bn} nt16={nt16} ring={ring}", (S, bn, nt16, ring)))
    rows.sort(key=lambda r: r[0])
    print(f"{name} N={N} K={K} ({wbytes / 1e6:.0f} MB):")
    shown = {"32": 0, "16": 0}
    for t, d, cfg in rows:
        k = d[:2]
        if shown[k] >= 3:
            continue
        shown[k] += 1
        chk = ""
        if cfg:
            S, bn, nt16, ring = cfg
            part.zero_()
            ops.gemm_parts(a, kts[0], part, S, bn, mb16, 16, nt16, ring)
            err = (part[:S * M * N].view(S, M, N).double().sum(0) - ref).abs().max().item()
            chk = f"max |err| vs fp64 {err:.2e} (|ref| max {ref.abs().max().item():.1f})"
        print(f"    {t:6.2f} us {wbytes / 1e6 / t:5.2f} TB/s  {d}  {chk}")
    sys.stdout.flush()
    del kts
    torch.cuda.empty_cache()
