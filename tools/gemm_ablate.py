"""Where does the skinny GEMM spend its time?  LADE_DEBUG=gemm_dbg=0 | 1 (no output stores) | 4 (no LDS reads / MFMA) | 5 (loads only).
Run on the GPU box: for d in 0 1 4 5; do LADE_DEBUG=gemm_dbg=$d python tools/gemm_ablate.py; done"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lookaheaddecoding_amd import cabi
from lookaheaddecoding_amd.cabi import call, ptr, dtype_code
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
M = int(os.environ.get('M', '60'))
MB = 2 if M <= 64 else 4
dbg = cabi.debug('gemm_dbg', '0')
for name, N, K, cfgs in (("qkv", 12288, 4096, [(MB, 128, 5), (MB, 256, 4), (MB, 192, 4)]), ("gate_up", 22016, 4096, [(MB, 128, 4), (MB, 256, 5), (MB, 192, 2)]), ("down", 4096, 11008, [(MB, 128, 8)])):
    a = torch.randn(M, K, device="cuda").bfloat16()
    ws = [torch.randn(N, K, device="cuda").bfloat16() * 0.02 for _ in range(max(1, int(600e6 / (N * K * 2))))]
    i = [0]
    for (mb, bn, S) in cfgs:
        part = torch.empty(S, M, N, dtype=torch.float32, device="cuda")
        def mine():
            i[0] = (i[0] + 1) % len(ws)
            call('lade_gemm_skinny', ptr(a), a.stride(0), ptr(ws[i[0]]), ws[i[0]].stride(0), None, 0, ptr(part), M, N, K, S, bn, mb, 0, 0, 0, 0, dtype_code(a))
        t = timeit(mine)
        print(f"dbg={dbg:>2s} {name:8s} bn={bn:3d} S={S:2d}  {t:7.2f} us  {N * K * 2 / t / 1e6:5.2f} TB/s", flush=True)
