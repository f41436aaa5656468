"""Reference points for the write roofline on this box: torch fill / copy of the same byte count as one
fe2s2_dropin step (1.03 GB), timed with events."""
import torch
n = 8192 * 7876 * 2  # doubles written per step (Hmat + comb)
x = torch.empty(n, dtype=torch.float64, device="cuda")
y = torch.empty(n, dtype=torch.float64, device="cuda")
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: x.fill_(1.5)); print(f"fill  {n*8/1e9:.3f} GB: {ms:.4f} ms  {n*8/ms/1e9:.2f} TB/s written")
ms = t(lambda: x.zero_()); print(f"zero  {n*8/1e9:.3f} GB: {ms:.4f} ms  {n*8/ms/1e9:.2f} TB/s written")
ms = t(lambda: y.copy_(x)); print(f"copy  {n*8/1e9:.3f} GB: {ms:.4f} ms  {2*n*8/ms/1e9:.2f} TB/s read+write")
