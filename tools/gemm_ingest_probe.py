"""What the activation re-reads cost the weight-streaming GEMM: the 7B / 13B step's launches with the activation traffic removed
(LADE_DEBUG=gemm_dbg bit 128: every activation piece re-reads one cached line - results are then meaningless), with and without the output
stores (bit 1) and the LDS-read / MFMA phase (bit 4).  K-tile-major weights, every launch on another layer's weights, 40 launches per hipGraph.
Run once per value (the env is read once per process):  for d in 0 128 1 129 5 133; do LADE_DEBUG=gemm_dbg=$d python tools/gemm_ingest_probe.py; done"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import cabi, ops

M = int(os.environ.get("M", "60"))
MODEL = os.environ.get("MODEL", "7b")
HID, INTER, QKV = {"7b": (4096, 11008, 12288), "13b": (5120, 13824, 15360)}[MODEL]
mb = (M + 31) // 32


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


line = []
# (name, N, K, S, bn, mt, nt, ring): the in-step tuner's usual choices
cfgs = {60: (("qkv", QKV, HID, 2, 96, 1, 1, 0), ("o", HID, HID, 4, 64, 1, 1, 0), ("gate_up", 2 * INTER, HID, 1, 96, 1, 1, 8), ("down", HID, INTER, 4, 64, 1, 1, 0)),
        120: (("qkv", QKV, HID, 3, 192, 1, 0, 3), ("o", HID, HID, 3, 64, 1, 0, 3), ("gate_up", 2 * INTER, HID, 1, 128, 2, 1, 3), ("down", HID, INTER, 6, 128, 2, 0, 5))}[M if M in (60, 120) else 60]
for name, N, K, S, bn, mt, nt, ring in cfgs:
    a = torch.randn(M, K, device="cuda").bfloat16()
    n_w = max(3, int(700e6 / (N * K * 2)))
    kts = [ops.to_ktile((torch.randn(N, K, device="cuda") * 0.02).bfloat16()) for _ in range(n_w)]
    if os.environ.get("ZERO") == "1":         # all-zero operands: same instructions, same bytes, less switching power (the guide's DVFS give-back check)
        a.zero_()
        for k in kts:
            k.zero_()
    part = torch.empty(16 * 128 * N, dtype=torch.float32, device="cuda")
    act = torch.empty(M, N // 2, dtype=torch.bfloat16, device="cuda")
    i = [0]

    def run():
        i[0] = (i[0] + 1) % n_w
        if S == 1:
            ops.gemm_swiglu(a, kts[i[0]], act, bn, mb, mt, nt, ring)
        else:
            ops.gemm_parts(a, kts[i[0]], part, S, bn, mb, mt, nt, ring)
    t = timeit(run)
    n_wg = ((N + bn - 1) // bn) * S
    line.append(f"{name} {t:6.2f} us ({N * K * 2 / 1e6 / t:4.2f} TB/s of W; A re-reads {n_wg * M * (K // S) * 2 / 1e6:.0f} MB over {n_wg} WGs)")
    del kts
    torch.cuda.empty_cache()
print(f"LADE_DEBUG=gemm_dbg={cabi.debug('gemm_dbg', '0'):>3s} {MODEL} M={M}: " + " | ".join(line), flush=True)
