"""Microbenchmark of the lookahead attention kernel (run on the GPU box):
python tools/attn_bench.py [--T 60 120] [--P 1024 2048 4096] [--splits 1 4 6 8 9 12 18]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import cabi, ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, nargs="+", default=[60, 120])
    ap.add_argument("--P", type=int, nargs="+", default=[1024, 4096])
    ap.add_argument("--splits", type=int, nargs="+", default=[0])
    ap.add_argument("--H", type=int, default=32)
    ap.add_argument("--Hkv", type=int, default=32)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--reps", type=int, default=400)
    ap.add_argument("--resident", action="store_true", help="one K/V cache, re-read every repetition (stays in the Infinity Cache)")
    ap.add_argument("--lp-rank", action="store_true", help="a lookahead-parallel rank's step instead of the full window: 4 re-fed inputs, "
                    "columns 12..14 of the W=15 window, 2 candidates (T = 31; with --H 64 --Hkv 8 this is config 5's rank shape)")
    ap.add_argument("--wg", type=int, default=0, help="work-group rows (lade_attn_args.wg_rows): 0 = default, 128 | 64 | 32")
    ap.add_argument("--smax", type=int, nargs="+", default=[0], help="cache capacity in rows (the stride of the V^T rows and of the K heads): 0 = just above P + T")
    ap.add_argument("--W", type=int, default=15)
    ap.add_argument("--N", type=int, default=5)
    a = ap.parse_args()
    W, N = a.W, a.N
    gs = N - 1
    for T in a.T:
        g = max(0, (T - (N - 1) * W) // gs)
        T = (N - 1) * W + g * gs
        for P, smax in ((P, sm) for P in a.P for sm in a.smax):
            S_max = max(smax, (P + T + 63) // 64 * 64) if smax else (P + T + 63) // 64 * 64 + 64
            q = torch.randn(T, (a.H + 2 * a.Hkv) * a.d, device="cuda").bfloat16()
            # enough distinct caches that their total exceeds the Infinity Cache (256 MB): every launch reads HBM; --resident keeps one
            per = 2 * a.Hkv * S_max * a.d * 2
            n_rot = 1 if a.resident else max(2, min(160, -(-600_000_000 // per)))
            k = [torch.randn(a.Hkv, S_max, a.d, device="cuda").bfloat16() for _ in range(n_rot)]
            vt = [torch.randn(a.Hkv, a.d, S_max, device="cuda").bfloat16() for _ in range(n_rot)]
            mask = ops.StepMask.from_levels(1, [W - 1] + [W] * (N - 2), g * gs, gs, P)
            if a.lp_rank:
                mask = ops.StepMask.from_levels(4, [13, 2, 2, 2], 8, gs, P)
                T = mask.T
                q = torch.randn(T, (a.H + 2 * a.Hkv) * a.d, device="cuda").bfloat16()
            alg = 2 * (2 * a.Hkv * (P + T) * a.d + 2 * a.H * T * a.d)
            for ns in a.splits:
                n = ns if ns > 0 else ops.choose_splits(a.H, a.H // a.Hkv, T, P + T)
                if cabi.debug("attn_dbg") == "16":
                    us, tl = ops.time_attn(q, k, vt, mask, H=a.H, Hkv=a.Hkv, d=a.d, n_splits=n, reps=a.reps, debug_timeline=True)
                    br = ops.attn_block_rows(a.H // a.Hkv, T) if not cabi.debug('attn_shape') else int(cabi.debug('attn_shape'))
                    nwg = a.Hkv * n * ((T * (a.H // a.Hkv) + br - 1) // br)
                    tl = tl[:nwg].double()
                    rel = tl[:, 1:8] - tl[:, :1]          # stamps 1..7 (7 = key-part states published, before the merge reads)
                    order = torch.argsort(tl[:, 0])
                    half = len(order) // 2
                    for nm, idx in (("first-started half", order[:half]), ("second half", order[half:])):
                        rr = rel[idx]
                        ok = (rr[:, 1] > 0)
                        print("   ", nm, "mean stamps:", [int(x) for x in rr[ok].mean(0).tolist()], "start offset mean:", int((tl[idx, 0] - tl[:, 0].min()).mean()))
                    print("   stamps (cycles since WG start) mean:", [int(x) for x in rel.mean(0).tolist()], "max:", [int(x) for x in rel.max(0)[0].tolist()],
                          " WG start spread:", int(tl[:, 0].max() - tl[:, 0].min()), " kernel span:", int(tl[:, 6].max() - tl[:, 0].min()))
                else:
                    us = ops.time_attn(q, k, vt, mask, H=a.H, Hkv=a.Hkv, d=a.d, n_splits=n, reps=a.reps, wg_rows=a.wg)
                print(f"H={a.H:3d}/{a.Hkv:2d} T={T:4d} P={P:5d} S_max={S_max:5d} splits={n:3d} wg={a.wg or 128:3d}  {us:8.2f} us   {alg / us / 1e3:8.1f} GB/s  ({alg / us / 1e3 / 8000 * 100:5.1f}% of 8 TB/s)", flush=True)


if __name__ == "__main__":
    main()
