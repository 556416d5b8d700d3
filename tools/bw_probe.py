"""HBM streaming ceilings on this box (read-only reduction, copy), for calibrating roofline fractions."""
import torch

def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

for mb in (64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device="cuda", dtype=torch.bfloat16)
    y = torch.empty_like(x)
    xf = x.view(torch.float32)
    s = t(lambda: torch.sum(xf))
    c = t(lambda: y.copy_(x))
    m = t(lambda: torch.max(xf))
    print(f"{mb:5d} MB: sum(read) {mb / 1024 / s / 1e3 * 1.0737:6.2f} TB/s   max(read) {mb / 1024 / m / 1e3 * 1.0737:6.2f} TB/s   copy(read+write) {2 * mb / 1024 / c / 1e3 * 1.0737:6.2f} TB/s")
