"""Row-major against K-tile-major weights on the skinny GEMM: the projections of a 7B / 13B layer over the autotuner's candidate space
(wave grids x weight rows per work-group x split counts), every launch on a different layer's weights (HBM, not the Infinity Cache),
40 dependent launches per hipGraph.  Prints, per projection, the best configuration of each layout and the same-configuration pairs, and
checks that the two layouts give bit-identical results.   python tools/gemm_ktile_probe.py [7b|13b] [M ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import ops

MODEL = sys.argv[1] if len(sys.argv) > 1 else "7b"
MS = [int(x) for x in sys.argv[2:]] or [60]
HID, INTER, QKV = {"7b": (4096, 11008, 12288), "13b": (5120, 13824, 15360), "70b": (8192, 28672, 10240)}[MODEL]
N_CU = torch.cuda.get_device_properties(0).multi_processor_count
SHAPES = {32: ((1, 1, 0, (64, 128, 256)), (1, 1, 2, (128, 256)), (2, 1, 0, (128,)), (1, 1, 1, (96,))),
          64: ((2, 1, 0, (128, 256)), (2, 2, 0, (128, 192, 256)), (2, 2, 2, (192, 256)), (2, 1, 1, (64, 96)), (2, 2, 1, (96,))),
          96: ((3, 1, 0, (64, 128, 192, 256)), (3, 3, 0, (128, 192, 256)), (3, 3, 2, (192, 256)), (4, 1, 0, (128, 192)), (4, 2, 0, (128, 192)), (3, 3, 1, (96,))),
          128: ((4, 1, 0, (64, 128, 192, 256)), (4, 2, 0, (128, 192, 256)), (4, 4, 0, (192, 256)), (4, 4, 2, (192, 256)), (4, 4, 1, (96,)), (4, 2, 1, (96,)))}


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


def candidates(N, K, mclass, swiglu):
    out = []
    if swiglu:
        mbs = mclass // 32
        for bn in (64, 96, 128) + ((224,) if N % 224 == 0 else ()):
            for mt in sorted({1, 2 if mbs % 2 == 0 else 1, mbs}):
                if mbs % mt == 0 and (mbs // mt) * (bn // 32) <= 8:
                    out.append((mbs, bn, 1, mt, 1))
        return out
    for mb, mt, nt, bns in SHAPES[mclass]:
        for bn in bns:
            nblk = (N + bn - 1) // bn
            for S in sorted({max(1, round(N_CU / nblk)), max(1, round(N_CU * 2 / nblk)), max(1, round(N_CU * 3 / nblk))}):
                if 2 <= S <= 16 and K // 64 >= 2 * S:
                    out.append((mb, bn, S, mt, nt))
    return out


_w = torch.randn(160, 320, device="cuda").bfloat16()
assert torch.equal(ops.to_ktile(_w), _w.view(160, 5, 64).permute(1, 0, 2).contiguous()), "lade_weight_to_ktile != the torch permutation"
_w = torch.randn(96, 256, device="cuda").bfloat16()[:, :192]                       # strided source
assert torch.equal(ops.to_ktile(_w), _w.reshape(96, 3, 64).permute(1, 0, 2).contiguous())
print("lade_weight_to_ktile == torch permutation", flush=True)

for M in MS:
    mclass = next(c for c in (32, 64, 96, 128) if M <= c)
    tot = {"row": 0.0, "kt": 0.0}
    for name, N, K, swiglu in (("qkv", QKV, HID, False), ("o", HID, HID, False), ("gate_up", 2 * INTER, HID, True), ("down", HID, INTER, False)):
        n_w = max(3, int(700e6 / (N * K * 2)))
        rows = [torch.randn(N, K, device="cuda").bfloat16() * 0.02 for _ in range(n_w)]
        kts = [ops.to_ktile(w) for w in rows]
        a = torch.randn(M, K, device="cuda").bfloat16()
        part = torch.empty(16 * 128 * N, dtype=torch.float32, device="cuda")
        act = torch.empty(M, N // 2, dtype=torch.bfloat16, device="cuda")
        res = {}
        for cfg in candidates(N, K, mclass, swiglu):
            mb, bn, S, mt, nt = cfg
            t = {}
            ok = True
            for lay, ws in (("row", rows), ("kt", kts)):
                i = [0]

                def run():
                    i[0] = (i[0] + 1) % len(ws)
                    if swiglu:
                        ops.gemm_swiglu(a, ws[i[0]], act, bn, mb, mt, nt)
                    else:
                        ops.gemm_parts(a, ws[i[0]], part, S, bn, mb, mt, nt)

                try:
                    t[lay] = timeit(run)
                except Exception as e:            # wave grid not built
                    ok = False
                    break
                # bit-identity of the two layouts on weight 0
                i[0] = -1
                run()
                snap = (act if swiglu else part[:S * M * N]).clone()
                if lay == "row":
                    ref = snap
                elif not torch.equal(ref, snap):
                    print(f"MISMATCH {name} {cfg}: row-major and K-tile-major results differ", flush=True)
            if ok:
                res[cfg] = t
        if not res:
            continue
        best_row = min(res.items(), key=lambda kv: kv[1]["row"])
        best_kt = min(res.items(), key=lambda kv: kv[1]["kt"])
        mb_w = N * K * 2 / 1e6
        print(f"{MODEL} M={M} {name:8s} N={N} K={K}: row-major best {best_row[1]['row']:6.2f} us {mb_w / best_row[1]['row']:5.2f} TB/s {best_row[0]} | "
              f"K-tile-major best {best_kt[1]['kt']:6.2f} us {mb_w / best_kt[1]['kt']:5.2f} TB/s {best_kt[0]}", flush=True)
        for cfg, t in sorted(res.items(), key=lambda kv: kv[1]["kt"])[:6]:
            print(f"      {str(cfg):24s} row {t['row']:6.2f}  kt {t['kt']:6.2f}  ({(t['kt'] / t['row'] - 1) * 100:+5.1f} %)", flush=True)
        tot["row"] += best_row[1]["row"]
        tot["kt"] += best_kt[1]["kt"]
        del rows, kts
        torch.cuda.empty_cache()
    print(f"{MODEL} M={M} layer sum: row-major {tot['row']:6.2f} us | K-tile-major {tot['kt']:6.2f} us  ({(tot['kt'] / tot['row'] - 1) * 100:+5.1f} %)", flush=True)
