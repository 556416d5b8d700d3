"""Round-6 structures of the skinny GEMM against the round-5 kernel, isolated (hipGraph of 40 launches over rotating layer weights, K-tile-major):
  * the 160-row class (mb = 5: shapes (1,5,n,nt), up to 256 weight rows per work-group) against the 192-row class a 129..160-row step padded to before,
  * the ping-pong K loop (ring = 10 + stages: csrc/gemm_pp.hpp) against the lock-step loop at the same work-group tile.
Every new configuration is CHECKED: the 160-row shapes bit for bit against the 192-row class at the same split count (same K slices, same
per-element chain), the ping-pong form against an fp64 reference of the product (its sum order differs: even tiles + odd tiles).
    MODEL=13b M=150 python tools/gemm_r6_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import cabi, ops

M = int(os.environ.get("M", "150"))
MODEL = os.environ.get("MODEL", "13b")
DT = {"bf16": torch.bfloat16, "f16": torch.float16}[os.environ.get("DTYPE", "bf16")]
HID, INTER, QKV = {"7b": (4096, 11008, 12288), "13b": (5120, 13824, 15360), "70b": (8192, 28672, 10240)}[MODEL]
PROJS = os.environ.get("PROJS", "qkv,o,gate_up,down").split(",")
mb0 = (M + 31) // 32


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


def lock_step(mb):
    """(mb, mt, nt, bn list) of the lock-step kernel per row class: the engine's candidate shapes"""
    return {1: ((1, 1, 0, (64, 128, 256)), (1, 1, 2, (128, 256))),
            2: ((2, 1, 0, (128, 256)), (2, 2, 0, (128, 192, 256)), (2, 1, 1, (64, 96)), (2, 2, 1, (96,))),
            3: ((3, 1, 0, (64, 128, 192)), (3, 3, 0, (128, 192, 256)), (3, 3, 2, (192, 256)), (3, 3, 1, (96,))),
            4: ((4, 1, 0, (64, 128, 192)), (4, 2, 0, (128, 192, 256)), (4, 4, 0, (192, 256)), (4, 4, 2, (192, 256)), (4, 2, 1, (96,))),
            5: ((5, 5, 1, (96, 128, 192, 256)), (5, 5, 2, (128, 256))),
            6: ((6, 3, 1, (64, 128)), (6, 3, 2, (128,)), (6, 2, 1, (64,)), (6, 2, 2, (128,))),
            8: ((8, 4, 1, (64, 128)), (8, 4, 2, (128,)), (8, 2, 1, (64,)), (8, 2, 2, (128,)))}[mb]


def ping_pong(mb):
    """(mb, mt, nt, bn) of the ping-pong kernel: a group is a (mb / mt) x (bn / 32 / nt) grid of FOUR waves"""
    return {1: ((1, 1, 1, 128), (1, 1, 2, 256)), 2: ((2, 1, 1, 64), (2, 1, 2, 128), (2, 1, 3, 192), (2, 2, 1, 128), (2, 2, 2, 256)),
            3: ((3, 3, 1, 128), (3, 3, 2, 256)), 4: ((4, 2, 1, 64), (4, 2, 2, 128), (4, 2, 3, 192), (4, 4, 1, 128), (4, 1, 4, 128)),
            5: ((5, 5, 1, 128), (5, 5, 2, 256)), 6: ((6, 3, 1, 64), (6, 3, 2, 128)), 8: ((8, 4, 1, 64), (8, 4, 2, 128))}[mb]


print(f"{MODEL} M={M} {DT}: us per launch, TB/s of weights", flush=True)
for name in PROJS:
    N, K = {"qkv": (QKV, HID), "o": (HID, HID), "gate_up": (2 * INTER, HID), "down": (HID, INTER)}[name]
    a = torch.randn(M, K, device="cuda").to(DT)
    n_w = max(3, int(700e6 / (N * K * 2)))
    kts = [ops.to_ktile((torch.randn(N, K, device="cuda") * 0.02).to(DT)) for _ in range(n_w)]
    w0 = ops.from_ktile(kts[0])
    ref = (a.double() @ w0.double().t())
    part = torch.empty(16 * 128 * N, dtype=torch.float32, device="cuda")
    part2 = torch.empty(16 * 128 * N, dtype=torch.float32, device="cuda")
    act = torch.empty(M, N // 2, dtype=DT, device="cuda")
    wbytes = N * K * 2
    i = [0]

    def rot():
        i[0] = (i[0] + 1) % n_w
        return kts[i[0]]

    def splits(bn):
        nblk = (N + bn - 1) // bn
        return sorted({S for S in {max(1, round(256 / nblk)), max(1, round(384 / nblk)), max(1, round(512 / nblk)), max(1, round(768 / nblk))}
                       if 2 <= S <= 16 and K // 64 >= 2 * S and S * M * N <= part.numel()})

    def check(S, bn, mb, mt, nt, ring, exact_against=None):
        part.zero_()
        ops.gemm_parts(a, kts[0], part, S, bn, mb, mt, nt, ring)
        got = part[:S * M * N].view(S, M, N)
        if exact_against is not None:
            part2.zero_()
            ops.gemm_parts(a, kts[0], part2, S, 128, exact_against)
            torch.cuda.synchronize()
            return "bit-identical" if torch.equal(part[:S * M * N], part2[:S * M * N]) else f"MISMATCH {int((part[:S * M * N] != part2[:S * M * N]).sum())}"
        err = (got.double().sum(0) - ref).abs().max().item()
        return f"max |err| vs fp64 {err:.2e} (|ref| max {ref.abs().max().item():.1f})"

    rows = []
    classes = sorted({mb0, 6 if mb0 == 5 else mb0})
    for mb in classes:
        for (mb_, mt, nt, bns) in lock_step(mb):
            for bn in bns:
                for S in splits(bn):
                    for ring in ((0,) if mb != mb0 or mb0 != 5 else (0, 2)):
                        try:
                            t = timeit(lambda: ops.gemm_parts(a, rot(), part, S, bn, mb, mt, nt, ring))
                        except cabi.LadeHipError:
                            continue
                        rows.append((t, f"lock-step {32 * mb}-row class  S={S} bn={bn} mt={mt} nt={nt} ring={ring}", (S, bn, mb, mt, nt, ring)))
    for (mb, mt, nt, bn) in ping_pong(mb0):
        for S in splits(bn):
            for ring in (10, 12, 14, 16):
                try:
                    t = timeit(lambda: ops.gemm_parts(a, rot(), part, S, bn, mb, mt, nt, ring))
                except cabi.LadeHipError:
                    continue
                rows.append((t, f"PING-PONG {32 * mb}-row class  S={S} bn={bn} mt={mt} nt={nt} ring={ring}", (S, bn, mb, mt, nt, ring)))
    rows.sort(key=lambda r: r[0])
    print(f"{name} N={N} K={K} ({wbytes / 1e6:.0f} MB):")
    shown = {"lock-step 192": 0, "lock-step 160": 0, "PING": 0, "lock-step": 0}
    for t, d, cfg in rows:
        key = next(k for k in shown if d.startswith(k))
        if shown[key] >= 3:
            continue
        shown[key] += 1
        S, bn, mb, mt, nt, ring = cfg
        chk = ""
        if d.startswith("PING"):
            chk = check(S, bn, mb, mt, nt, ring)
        elif mb == 5:
            chk = check(S, bn, mb, mt, nt, ring, exact_against=6)
        print(f"    {t:6.2f} us {wbytes / 1e6 / t:5.2f} TB/s  {d}  {chk}")
    if name == "gate_up":
        gu = []
        for mb in classes:
            for bn in (64, 96, 128):
                for mt in sorted({1, 2 if mb % 2 == 0 else 1, mb if mb <= 5 else mb // 2}):
                    for ring in (0, 3, 8):
                        try:
                            t = timeit(lambda: ops.gemm_swiglu(a, rot(), act, bn, mb, mt, 1, ring))
                        except cabi.LadeHipError:
                            continue
                        gu.append((t, f"unsplit + SwiGLU epilogue {32 * mb}-row class bn={bn} mt={mt} ring={ring}"))
        gu.sort()
        seen = {}
        for t, d in gu:
            k = d.split("class")[0]
            if seen.get(k, 0) < 3:
                seen[k] = seen.get(k, 0) + 1
                print(f"    {t:6.2f} us {wbytes / 1e6 / t:5.2f} TB/s  {d}")
    sys.stdout.flush()
    del kts
    torch.cuda.empty_cache()
