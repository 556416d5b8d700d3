"""What clock and power the chip sustains under the weight-streaming GEMM at different row classes ("the MFMA phase costs
clock" - is the 128-row class running into the power budget?).  A hipGraph of 40 gate/up launches (13B width, rotating weights) is replayed for
SECONDS per case while a thread samples `amd-smi metric --clock --power` (falls back to `rocm-smi`); prints us per launch, the sampled
socket power and GFX clock per case.  Cases: M rows x {random, all-zero operands} x {ring kernel, RA kernel}."""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from lookaheaddecoding_amd import ops

SECONDS = float(os.environ.get("SECONDS_PER_CASE", "4"))
HID, INTER = 5120, 13824
N, K = 2 * INTER, HID
DT = torch.bfloat16


from clock_probe_lib import sample  # noqa: E402


def raw_once():
    for cmd in (["amd-smi", "metric", "-g", "0", "--power", "--clock"], ["rocm-smi", "--showpower", "--showclocks"]):
        try:
            print("$", " ".join(cmd))
            print(subprocess.run(cmd, capture_output=True, text=True, timeout=20).stdout[-3000:], flush=True)
        except Exception as e:          # noqa: BLE001
            print("   failed:", e)


kts = [ops.to_ktile((torch.randn(N, K, device="cuda") * 0.02).to(DT)) for _ in range(4)]
kz = [torch.zeros_like(k) for k in kts[:2]]
part = torch.empty(4 * 128 * N, dtype=torch.float32, device="cuda")
print("idle:", sample(), flush=True)
CASES = ((60, False, "ring"), (120, False, "ring"), (120, True, "ring"), (60, False, "ra"), (120, False, "ra"), (120, True, "ra"), (1, False, "ring"))
if os.environ.get("CASES"):            # CASES=76,100,120: random operands on the ring kernel at these row counts (LADE_DEBUG=gemm_dbg=256: zero padding rows)
    CASES = tuple((int(m), False, "ring") for m in os.environ["CASES"].split(","))
    print("LADE_DEBUG =", os.environ.get("LADE_DEBUG", ""))
for M, zero, kind in CASES:
    if kind == "ra" and not __import__("lookaheaddecoding_amd.cabi", fromlist=["x"]).experimental():
        continue                      # the RA kernel lives in the EXPERIMENTAL build only
    a = torch.zeros(M, K, device="cuda", dtype=DT) if zero else torch.randn(M, K, device="cuda").to(DT)
    ws = kz if zero else kts
    act = torch.empty(M, N // 2, dtype=DT, device="cuda")
    mb = (M + 31) // 32
    i = [0]

    def run():
        i[0] = (i[0] + 1) % len(ws)
        if kind == "ring":
            ops.gemm_swiglu(a, ws[i[0]], act, 128 if mb == 4 else 96, mb, 2 if mb == 4 else (mb if mb in (3, 5) else 1), 1, 3 if mb == 4 else (8 if mb < 3 else 0))
        else:
            ops.gemm_ra_parts(a, ws[i[0]], part, 4, 4)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(40):
            run()
    samples, stop = [], threading.Event()

    def sampler():
        time.sleep(1.0)
        while not stop.is_set():
            samples.append(sample())
            time.sleep(0.3)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < SECONDS:
        for _ in range(20):
            g.replay()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    us = e0.elapsed_time(e1) * 1e3 / (n * 40)
    good = [s for s in samples if s and s[0] not in (None, "ERR")]
    pw = sum(s[0] for s in good) / len(good) if good else float("nan")
    ck = [s[1] for s in samples if s and isinstance(s[1], float)]
    ckm = [s[2] for s in samples if s and isinstance(s[2], float)]
    print(f"gate/up 13B M={M:3d} {'zero  ' if zero else 'random'} {kind:4s}: {us:7.2f} us per launch ({N * K * 2 / 1e6 / us:4.2f} TB/s)  power {pw:6.1f} W  gfx clock mean {sum(ck) / len(ck) if ck else float('nan'):7.1f} "
          f"max {max(ckm) if ckm else float('nan'):7.1f} MHz  ({len(samples)} samples{'' if good else '; first: ' + str(samples[:1])})", flush=True)
if not os.environ.get("CASES"):
    raw_once()
