"""lade_greedy_post_step alone: microseconds per launch in a hipGraph of 64 launches, steady phase, cold regime (random window tokens
over a 32000-token pool that holds a few n-grams) and hot regime (few keys, full slots).  Optionally against another build of the
library (`--old path/to/liblade_hip.so` with the round-3 signature: no record_host argument).
    python tools/post_step_bench.py [--old _ab_old/lookaheaddecoding_amd/liblade_hip.so]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import cabi

V, W, N, G = 32000, 15, 5, 15
gs = N - 1
wcap = W + N - 3
dev = "cuda"
libs = {"this": (cabi.load_library(), True)}
if "--old" in sys.argv:
    libs["old"] = (C.CDLL(os.path.abspath(sys.argv[sys.argv.index("--old") + 1])), False)


def bench(lib, has_host, vocab_used, label):
    g = torch.Generator(device=dev).manual_seed(1)
    i32 = dict(dtype=torch.int32, device=dev)
    ctl = torch.zeros(64, **i32)
    ctl[0] = 2048; ctl[1] = 7; ctl[2] = 2047; ctl[3] = 1; ctl[5] = N - 2
    ctl[32] = W - 1
    ctl[33:33 + N - 2] = W
    window = torch.randint(0, vocab_used, (N - 1, wcap), generator=g, **i32)
    pool_tok = torch.randint(0, vocab_used, (V, G, gs), generator=g, **i32)
    pool_cnt = torch.randint(0, G + 1, (V,), generator=g, **i32) if vocab_used < 1000 else (torch.rand(V, device=dev, generator=g) < 0.02).to(torch.int32)
    am = torch.randint(0, vocab_used, (1 + W + G * gs,), generator=g, **i32)
    guess = torch.zeros(G * gs, **i32)
    tail = torch.zeros(N + 2, **i32)
    record = torch.zeros(24, **i32)
    rec_host = torch.zeros(24, dtype=torch.int32).pin_memory()
    T = (N - 1) * W
    st = torch.cuda.current_stream().cuda_stream
    args = [ctl.data_ptr(), window.data_ptr(), wcap, pool_tok.data_ptr(), pool_cnt.data_ptr(), V, W, N, G, am.data_ptr(), W, guess.data_ptr(), T, 0, 2, 0,
            tail.data_ptr(), -1, None, None, record.data_ptr()]
    fn = lib.lade_greedy_post_step
    fn.restype = C.c_int
    vp, i = C.c_void_p, C.c_int32
    fn.argtypes = [vp, vp, i, vp, vp, i, i, i, i, vp, i, vp, i, i, i, i, vp, i, vp, vp, vp] + ([vp] if has_host else []) + [vp]

    def launch():
        a = args + ([rec_host.data_ptr() if '--no-host' not in sys.argv else None] if has_host else [])
        rc = fn(*a, torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
        assert rc == 0, rc
        # a new random level every launch would need a kernel; rotate the window's level 0 instead so that the keys change
        window[0].copy_(torch.roll(window[0], 1))

    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(64):
            launch()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 64)
    # the roll kernel alone, to subtract
    gr2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr2):
        for _ in range(64):
            window[0].copy_(torch.roll(window[0], 1))
    b2 = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr2.replay(); e1.record(); torch.cuda.synchronize()
        b2 = min(b2, e0.elapsed_time(e1) * 1e3 / 64)
    print(f"{label:28s} {best - b2:6.2f} us per post-step launch (pair {best:.2f}, window rotation alone {b2:.2f})", flush=True)


for name, (lib, has_host) in libs.items():
    bench(lib, has_host, V, f"{name}: cold (random keys)")
    bench(lib, has_host, 16, f"{name}: hot (16 keys, full)")
