"""Times the glue kernels of one layer in isolation (run on the GPU box): python tools/glue_bench.py [--T 60]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import ops


def timeit(fn, reps=100):
    """mean time of one launch inside a hipGraph of `reps` dependent launches (what a decode step pays)"""
    for _ in range(3):
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
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--T", type=int, default=60)
    a = ap.parse_args()
    T, hid, inter, H, d, S_max = a.T, 4096, 11008, 32, 128, 2304
    dt, dev = torch.bfloat16, "cuda"
    x = torch.randn(T, hid, device=dev).to(dt)
    r = torch.randn(T, hid, device=dev).to(dt)
    w = torch.ones(hid, device=dev).to(dt)
    h = torch.empty_like(x)
    part = torch.randn(16 * T * 2 * inter, device=dev)
    gu = torch.randn(T, 2 * inter, device=dev).to(dt)
    act = torch.empty(T, inter, device=dev).to(dt)
    qkv = torch.randn(T, 3 * hid, device=dev).to(dt)
    qb = torch.empty(T, hid, device=dev).to(dt)
    pos = torch.arange(100, 100 + T, dtype=torch.int32, device=dev)
    cos = torch.randn(4096, d, device=dev).to(dt)
    sin = torch.randn(4096, d, device=dev).to(dt)
    kc = torch.zeros(H, S_max, d, device=dev).to(dt)
    vc = torch.zeros(H, d, S_max, device=dev).to(dt)
    print(f"T={T}")
    print(f"  rmsnorm                 {timeit(lambda: ops.rmsnorm(x, w, 1e-5, out=h)):7.2f} us")
    print(f"  add_rmsnorm             {timeit(lambda: ops.add_rmsnorm(x, r, w, 1e-5, out=h)):7.2f} us")
    for n in (2, 4, 8):
        print(f"  add_rmsnorm_parts n={n}   {timeit(lambda: ops.add_rmsnorm_parts(x, part, n, w, 1e-5, out=h)):7.2f} us")
    print(f"  silu_mul                {timeit(lambda: ops.silu_mul(gu, out=act)):7.2f} us")
    for n in (2, 3, 4):
        print(f"  silu_mul_parts n={n}      {timeit(lambda: ops.silu_mul_parts(part, n, T, inter, out=act)):7.2f} us")
    print(f"  rope_kv_append          {timeit(lambda: ops.rope_kv_append(qkv, pos, cos, sin, kc, vc, 500, H=H, Hkv=H, d=d)):7.2f} us")
    for n in (2, 4, 5, 8):
        print(f"  rope_kv_append_parts n={n} {timeit(lambda: ops.rope_kv_append_parts(part, n, qb, pos, cos, sin, kc, vc, T, 500, H=H, Hkv=H, d=d)):7.2f} us")
    nothing = torch.zeros(1, device=dev)
    print(f"  (torch tiny add_)       {timeit(lambda: nothing.add_(1)):7.2f} us")
    idx = torch.arange(T, dtype=torch.int32, device=dev)
    print(f"  gather_rows             {timeit(lambda: ops.gather_rows(x, idx, out=h, rows=T)):7.2f} us")


if __name__ == "__main__":
    main()
