"""sample_update (gfmc/walker.py:260-279) on a [8192, 7876] Green's-function matrix: fused kernel vs the torch passes."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pynqs_amd import gfmc
n, m = 8192, 7876
d = torch.device("cuda")
g = torch.Generator(device=d).manual_seed(1)
gk = torch.rand(n, m, generator=g, dtype=torch.float64, device=d)
gk[gk < 0.7] = 0.0
gk[:, 0] += 1.0
comb = torch.randint(0, 256, (n, m, 8), generator=g, dtype=torch.uint8, device=d)
u = torch.rand(n, 1, generator=g, dtype=torch.float64, device=d)
w = torch.ones(n, dtype=torch.float64, device=d)
for fused in (True, False):
    gfmc.FUSED_SAMPLE = fused
    for _ in range(3):
        r = gfmc.sample_update(None, w, comb, gk, u)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        r = gfmc.sample_update(None, w, comb, gk, u)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"fused={fused}: {ms:.3f} ms per step ({n * m * 8 / ms / 1e6:.0f} GB/s of the Green matrix), accepted {r[3]}")
