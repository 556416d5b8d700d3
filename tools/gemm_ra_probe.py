"""The register-resident-activation GEMM (lade_gemm_ra_kt, csrc/gemm_ra.hpp) against the LDS-ring kernel (lade_gemm_skinny_kt), isolated:
the four projections of a 7B / 13B layer at M rows, K-tile-major weights, every launch on another layer's weights, 40 launches per hipGraph.
Per projection: the ring kernel's best of a short list (the configurations the engine's tuner usually picks), then the RA kernel for every
admissible split count x chunk size, each CHECKED bit for bit against the ring kernel's partials at the same split count; gate/up also as
RA split-K + lade_silu_mul_parts against the ring kernel's unsplit SwiGLU epilogue.
    MODEL=13b M=120 python tools/gemm_ra_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import cabi, ops

M = int(os.environ.get("M", "120"))
MODEL = os.environ.get("MODEL", "13b")
DT = {"bf16": torch.bfloat16, "f16": torch.float16}[os.environ.get("DTYPE", "bf16")]
HID, INTER, QKV = {"7b": (4096, 11008, 12288), "13b": (5120, 13824, 15360), "70b": (8192, 28672, 10240)}[MODEL]
mb = (M + 31) // 32
CHECK = os.environ.get("CHECK", "1") != "0"


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


# ring-kernel short lists (mb is set by M): (S, bn, mt, nt, ring)
OLD = {1: [(2, 96, 1, 1, 0), (4, 64, 1, 1, 0), (4, 128, 1, 0, 0), (2, 128, 1, 0, 0), (3, 64, 1, 0, 0), (8, 128, 1, 0, 0)],
       2: [(2, 96, 1, 1, 0), (4, 64, 1, 1, 0), (4, 128, 1, 0, 0), (2, 128, 1, 0, 0), (3, 64, 1, 0, 0), (8, 128, 1, 0, 0), (4, 128, 2, 0, 0), (6, 128, 2, 0, 0)],
       3: [(3, 192, 1, 0, 3), (3, 64, 1, 0, 3), (2, 128, 1, 0, 0), (4, 128, 1, 0, 0), (6, 128, 3, 0, 0), (4, 192, 3, 0, 0), (8, 128, 1, 0, 0)],
       4: [(3, 192, 1, 0, 3), (3, 64, 1, 0, 3), (2, 128, 2, 0, 4), (6, 128, 2, 0, 4), (6, 128, 2, 0, 5), (4, 256, 2, 2, 3), (9, 192, 2, 2, 3), (4, 128, 2, 0, 4),
           (8, 128, 2, 0, 4)]}[mb]
OLD_GU = {1: [(64, 1, 1, 0), (96, 1, 1, 0), (128, 1, 1, 0)], 2: [(96, 1, 1, 8), (96, 2, 1, 8), (64, 1, 1, 0), (128, 2, 1, 0)],
          3: [(96, 3, 1, 0), (128, 3, 1, 0), (64, 1, 1, 0), (128, 1, 1, 0)], 4: [(128, 2, 1, 3), (128, 2, 1, 4), (128, 1, 1, 4), (128, 4, 1, 0), (96, 4, 1, 0)]}[mb]

print(f"{MODEL} M={M} {DT}: ring kernel (lade_gemm_skinny_kt) vs register-resident activations (lade_gemm_ra_kt); us per launch, TB/s of weights", flush=True)
for name, N, K in (("qkv", QKV, HID), ("o", HID, HID), ("gate_up", 2 * INTER, HID), ("down", HID, INTER)):
    a = torch.randn(M, K, device="cuda").to(DT)
    n_w = max(3, int(700e6 / (N * K * 2)))
    kts = [ops.to_ktile((torch.randn(N, K, device="cuda") * 0.02).to(DT)) for _ in range(n_w)]
    part = torch.empty(16 * 128 * N, dtype=torch.float32, device="cuda")
    part2 = torch.empty(16 * 128 * N, dtype=torch.float32, device="cuda")
    act = torch.empty(M, N // 2, dtype=DT, device="cuda")
    wbytes = N * K * 2
    i = [0]

    def rot():
        i[0] = (i[0] + 1) % n_w
        return kts[i[0]]

    old = []
    for (S, bn, mt, nt, ring) in OLD:
        if S * M * N > part.numel() or K // 64 < 2 * S:
            continue
        try:
            t = timeit(lambda: ops.gemm_parts(a, rot(), part, S, bn, mb, mt, nt, ring))
        except cabi.LadeHipError:
            continue
        old.append((t, f"S={S} bn={bn} mt={mt} nt={nt} ring={ring}"))
    old.sort()
    gu_old = []
    if name == "gate_up":
        for (bn, mt, nt, ring) in OLD_GU:
            try:
                t = timeit(lambda: ops.gemm_swiglu(a, rot(), act, bn, mb, mt, nt, ring))
            except cabi.LadeHipError:
                continue
            gu_old.append((t, f"unsplit + SwiGLU epilogue bn={bn} mt={mt} nt={nt} ring={ring}"))
        gu_old.sort()
    k_tiles = K // 64
    s_min = (k_tiles + ops.RA_KT - 1) // ops.RA_KT
    ra = []
    for S in range(s_min, s_min + 4):
        tps = (k_tiles + S - 1) // S
        if (S - 1) * tps >= k_tiles or S * M * N > part.numel():
            continue
        for cs in (4, 2):
            if N % (32 * cs):
                continue
            try:
                t = timeit(lambda: ops.gemm_ra_parts(a, rot(), part, S, cs))
            except cabi.LadeHipError as e:
                print("   ", e)
                continue
            ok = "unchecked"
            if CHECK:
                part.zero_()
                part2.zero_()
                ops.gemm_ra_parts(a, kts[0], part, S, cs)
                ops.gemm_parts(a, kts[0], part2, S, 128 if N % 128 == 0 else 64, mb)
                torch.cuda.synchronize()
                n = S * M * N
                ok = "bit-identical" if torch.equal(part[:n], part2[:n]) else f"MISMATCH ({int((part[:n] != part2[:n]).sum())} of {n})"
            extra = ""
            if name == "gate_up":
                t2 = timeit(lambda: (ops.gemm_ra_parts(a, rot(), part, S, cs), ops.silu_mul_parts(part, S, M, N // 2, out=act, layout=1)))
                extra = f"  + silu_mul_parts: {t2:6.2f} us"
            nc = N // (32 * cs)
            grp = max(1, min(256 // S, nc))
            ra.append((t, f"S={S} cs={cs} tiles/WG={tps} WGs={grp * S} chunks/WG={nc / grp:.2f} {ok}{extra}"))
    ra.sort()
    print(f"{name} N={N} K={K} ({wbytes / 1e6:.0f} MB):")
    for t, d in old[:3]:
        print(f"    ring  {t:6.2f} us {wbytes / 1e6 / t:5.2f} TB/s  {d}")
    for t, d in gu_old[:2]:
        print(f"    ring  {t:6.2f} us {wbytes / 1e6 / t:5.2f} TB/s  {d}")
    for t, d in ra:
        print(f"    RA    {t:6.2f} us {wbytes / 1e6 / t:5.2f} TB/s  {d}")
    sys.stdout.flush()
    del kts
    torch.cuda.empty_cache()
