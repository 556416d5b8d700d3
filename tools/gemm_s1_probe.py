import os, sys, torch
sys.path.insert(0, "/root/repo")
from lookaheaddecoding_amd.cabi import call, dtype_code, ptr
M = 60
def timeit(fn, reps=40, rounds=5):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best
for name, N, K in (("qkv", 12288, 4096), ("gate_up", 22016, 4096)):
    a = torch.randn(M, K, device="cuda").bfloat16()
    ws = [torch.randn(N, K, device="cuda").bfloat16() * 0.02 for _ in range(max(2, int(600e6 / (N * K * 2))))]
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    part = torch.empty(8, M, N, dtype=torch.float32, device="cuda")
    i = [0]
    for (S, bn, mb, mt, nt) in ((4, 192, 2, 2, 0), (1, 96, 2, 1, 1), (1, 96, 2, 2, 1), (2, 96, 2, 1, 1), (2, 96, 2, 2, 1), (3, 96, 2, 2, 1), (1, 128, 2, 1, 0), (2, 128, 2, 2, 0), (2, 192, 2, 2, 0)):
        def mine():
            i[0] = (i[0] + 1) % len(ws)
            call("lade_gemm_skinny", ptr(a), a.stride(0), ptr(ws[i[0]]), ws[i[0]].stride(0), ptr(out), out.stride(0), ptr(part), M, N, K, S, bn, mb, mt, nt, 0, 0, dtype_code(a))
        try:
            t = timeit(mine)
            print(f"{name} S={S} bn={bn} mb={mb} mt={mt}: {t:6.2f} us {N*K*2/t/1e6:5.2f} TB/s", flush=True)
        except Exception as e:
            print(name, S, bn, mb, mt, "ERR", str(e)[:80])
