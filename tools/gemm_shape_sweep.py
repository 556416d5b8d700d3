"""Every built work-group shape x split count x ring depth of the skinny GEMM on one projection, isolated (hipGraph of 40 launches over
rotating layer weights, K-tile-major): is there a shape the engine's candidate list does not contain that wins?
  MODEL=13b M=120 PROJ=qkv python tools/gemm_shape_sweep.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import cabi, ops

M = int(os.environ.get("M", "120"))
MODEL = os.environ.get("MODEL", "13b")
PROJ = os.environ.get("PROJ", "qkv")
HID, INTER, QKV = {"7b": (4096, 11008, 12288), "13b": (5120, 13824, 15360), "70b": (8192, 28672, 10240)}[MODEL]
N, K = {"qkv": (QKV, HID), "o": (HID, HID), "gate_up": (2 * INTER, HID), "down": (HID, INTER)}[PROJ]
mb = (M + 31) // 32
# (MW, MT, NG, NT) of gemm_kernel.hpp's table
ALL = [(1,1,1,1),(1,1,2,1),(1,1,4,1),(1,1,8,1),(1,1,2,2),(1,1,4,2),(1,1,3,1),(2,1,1,1),(2,1,2,1),(2,1,4,1),(2,1,3,2),(2,1,4,2),(2,1,3,1),(1,2,3,1),
       (1,2,2,1),(1,2,4,1),(1,2,6,1),(1,2,8,1),(1,2,2,2),(1,2,3,2),(1,2,4,2),(1,2,2,3),(1,2,2,4),(3,1,1,1),(3,1,2,1),(3,1,2,2),(3,1,2,3),(3,1,2,4),(1,3,3,1),
       (1,3,4,1),(1,3,6,1),(1,3,8,1),(1,3,3,2),(1,3,4,2),(4,1,1,1),(4,1,2,1),(4,1,2,2),(4,1,2,3),(4,1,2,4),(1,4,3,1),(2,2,3,1),(2,2,2,1),(2,2,4,1),(2,2,3,2),
       (2,2,4,2),(2,2,2,2),(1,4,4,1),(1,4,6,1),(1,4,8,1),(1,4,3,2),(1,4,4,2),(1,4,2,2),(1,4,7,1)]
shapes = [s for s in ALL if s[0] * s[1] == mb]


def timeit(fn, reps=40, rounds=4):
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
n_w = max(3, int(700e6 / (N * K * 2)))
kts = [ops.to_ktile((torch.randn(N, K, device="cuda") * 0.02).bfloat16()) for _ in range(n_w)]
part = torch.empty(16 * 128 * N, dtype=torch.float32, device="cuda")
act = torch.empty(M, N // 2, dtype=torch.bfloat16, device="cuda")
res = []
i = [0]
for (MW, MT, NG, NT) in shapes:
    bn = 32 * NG * NT
    nblk = (N + bn - 1) // bn
    Ss = sorted({max(1, round(256 / nblk)), max(1, round(384 / nblk)), max(1, round(512 / nblk)), max(1, round(768 / nblk))})
    if PROJ == "gate_up":
        Ss = [1]
    stage = (bn + 32 * mb) * 128
    for S in Ss:
        if S > 1 and (K // 64 < 2 * S or S * M * N > part.numel()):
            continue
        for ring in (3, 4, 5, 6, 8):
            if ring * stage > 160 * 1024:
                continue

            def run():
                i[0] = (i[0] + 1) % n_w
                if PROJ == "gate_up":
                    ops.gemm_swiglu(a, kts[i[0]], act, bn, mb, MT, NT, ring)
                else:
                    ops.gemm_parts(a, kts[i[0]], part, S, bn, mb, MT, NT, ring)
            try:
                t = timeit(run)
            except cabi.LadeHipError as e:
                continue
            res.append((t, (MW, MT, NG, NT), bn, S, ring, nblk * S))
res.sort()
print(f"{MODEL} M={M} {PROJ} N={N} K={K}: {len(res)} configurations, weights {N * K * 2 / 1e6:.0f} MB")
for t, shp, bn, S, ring, wgs in res[:14]:
    print(f"  {t:7.2f} us  {N * K * 2 / 1e6 / t:4.2f} TB/s  shape <MW,MT,NG,NT>={shp} bn={bn} S={S} ring={ring} wgs={wgs} partials {S * M * N * 4 / 1e6 if S > 1 else 0:.1f} MB")
best_by_shape = {}
for r in res:
    best_by_shape.setdefault(r[1], r)
print("  best per shape: " + " | ".join(f"{k}: {v[0]:.1f} (S={v[3]}, ring={v[4]})" for k, v in sorted(best_by_shape.items(), key=lambda kv: kv[1][0])))
